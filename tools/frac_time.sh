# usage (GPU box): bash tools/frac_time.sh -- kernel time of the fused half/quarter refinement (16x16 PUs of a 3840x2160 picture, 9 + 9 SATD candidates)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_frac
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_frac -o t -- python3 tools/run_stage.py --only frac --reps 6 > gpurun_out/frac_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_frac/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frac" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-30:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 3:]
    print("%-32s calls %3d avg %8.1f us min %8.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
