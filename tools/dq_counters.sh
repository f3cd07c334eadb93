# usage (GPU box): bash tools/dq_counters.sh -- instruction counters of depquant_kernel on the 8x8 / 16x16 / 32x32 tilings of tools/n13_time.py (instructions per wave and scan step)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_dq
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmc_dq -o p -- python3 tools/n13_time.py > gpurun_out/pmc_dq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_dq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "depquant_kernel" in r["Kernel_Name"]:
            rows[(int(r["Dispatch_Id"]), int(r["Grid_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for (d, g) in sorted(rows):
    if g in seen: continue
    seen.add(g)
    v = rows[(d, g)]
    w = max(1.0, v.get("SQ_WAVES", 1))
    print("grid %8d threads: waves %6d  per wave: VALU %8.0f SALU %8.0f LDS %7.0f SMEM %6.0f VMEM rd %6.0f wr %6.0f" % (g, w, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w, v.get("SQ_INSTS_SMEM", 0) / w, v.get("SQ_INSTS_VMEM_RD", 0) / w, v.get("SQ_INSTS_VMEM_WR", 0) / w))
PY
