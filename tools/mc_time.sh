# usage (GPU box): bash tools/mc_time.sh -- kernel times of the motion-compensation launches of a 3840x2160 picture
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_mc
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mc -o t -- python3 tools/run_stage.py --only mc --reps 6 > gpurun_out/mc_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_mc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0].replace("_ZN12_GLOBAL__N_1","")[:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 3:]
    print("%-42s calls %3d avg %8.1f us min %8.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
