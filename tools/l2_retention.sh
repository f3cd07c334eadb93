# usage (GPU box): bash tools/l2_retention.sh -- does a picture survive in L2 from one launch to the next?  The same kernel (ALF classification) six
# times back to back on the same plane, three plane sizes (16.6 / 4.1 / 1.0 MB): L2 hit / miss counters per launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_l2 -o p -- python3 tools/l2_retention.py > gpurun_out/pmc_l2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "alf_classify" in r["Kernel_Name"]:
            rows[(int(r["Dispatch_Id"]), int(r["Grid_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    v = rows[k]
    print("dispatch %4d grid %8d: L2 hits %9.0f misses %9.0f (miss fraction %.3f), fabric read requests %9.0f" % (k[0], k[1], v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0), v.get("TCC_MISS_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)), v.get("TCC_EA0_RDREQ_sum", 0)))
PY
