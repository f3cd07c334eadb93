# usage (GPU box): bash tools/ktime_mix.sh [samples] -- rocprofv3 kernel times of the batch entry points on the call mix of a real encode (tools/shape_mix_time.py)
N=${1:-2097152}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_mix
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mix -o t -- python3 tools/shape_mix_time.py $N > gpurun_out/mix_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_mix/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-72s calls %4d median %8.1f us min %8.1f max %8.1f" % (k, len(v), v[len(v) // 2], v[0], v[-1]))
PY
