# usage (GPU box): bash tools/ktime.sh STAGE SUBSTR -- kernel times (rocprofv3 kernel trace) of the launches of one stage of a 3840x2160 picture
# whose kernel names contain SUBSTR, e.g.  bash tools/ktime.sh resi rc_
STAGE=$1; SUB=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_kt -o t -- python3 tools/run_stage.py --only $STAGE --reps 6 > gpurun_out/kt_run.log 2>&1
SUB=$SUB python3 - <<'PY'
import csv, glob, collections, os
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["SUB"] in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0].replace("_ZN12_GLOBAL__N_1", "").replace("void ", "").replace("anonymous namespace)::", "")[:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 3:]
    print("%-42s calls %3d avg %8.1f us min %8.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
