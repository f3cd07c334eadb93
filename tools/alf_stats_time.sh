# usage (GPU box): bash tools/alf_stats_time.sh -- kernel time of the picture-level ALF covariance launch at 3840x2160 (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_alfs
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_alfs -o t -- python3 tools/run_stage.py --only alf_stats --reps 6 > gpurun_out/alfs.log 2>&1
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_alfs/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "alf_stats" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 3:]
    print("%-42s calls %3d avg %8.1f us min %8.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
