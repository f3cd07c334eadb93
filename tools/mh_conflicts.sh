# usage (GPU box): bash tools/mh_conflicts.sh -- LDS bank-conflict counters of me_hier_kernel with and without the +-4 grid's dense lanes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_mh
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_mh -o p -- python3 tools/mehier_time.py > gpurun_out/pmc_mh.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_mh/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "me_hier" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(rows):
    v = rows[d]
    print(d, {k: int(x) for k, x in v.items()}, "ratio %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_ACTIVE_INST_LDS", 1))))
PY
