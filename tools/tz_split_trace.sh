# usage (GPU box): bash tools/tz_split_trace.sh -- kernel durations of the split TZ search (first launch, org packing, raster, second launch) per PU size
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_tzs
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tzs -o t -- python3 tools/tz_time.py --split --cpu-sample 0 --reps 4 > gpurun_out/tzs.log 2>&1
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_tzs/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("::")[-1][:36]
        g = r.get("Grid_Size_X", r.get("Grid_Size", ""))
        d[(n, g, r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "tz_" in k[0] or "raster" in k[0] or "pack" in k[0]:
        print("%-38s grid %9s wg %5s calls %3d avg %8.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
