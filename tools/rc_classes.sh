# usage (GPU box): bash tools/rc_classes.sh [pmc]  -- per-class kernel time (and, with `pmc`, SQ counters) of rc_chain_kernel, see tools/rc_classes.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_rcc gpurun_out/pmc_rcc1 gpurun_out/pmc_rcc2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_rcc -o t -- python3 tools/rc_classes.py 4 > gpurun_out/rcc_run.log 2>&1
if [ "$1" = "pmc" ]; then
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_rcc1 -o p -- python3 tools/rc_classes.py 2 > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d gpurun_out/pmc_rcc2 -o p -- python3 tools/rc_classes.py 2 > /dev/null 2>&1
fi
python3 - <<'PY'
import csv, glob, collections
SIZES = [64, 32, 16, 8, 4]
def rows(pat):
    out = []
    for f in glob.glob(pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
tr = [r for r in rows("gpurun_out/prof_rcc/**/*kernel_trace.csv") if "rc_chain_kernel" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
t = collections.defaultdict(list)
for i, r in enumerate(tr):
    if i >= len(SIZES):
        t[SIZES[i % len(SIZES)]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("rc_chain_kernel per class (1.66 M samples each), us:", {s: round(sum(v) / len(v), 1) for s, v in t.items()}, "sum", round(sum(sum(v) / len(v) for v in t.values()), 1))
for d in ("pmc_rcc1", "pmc_rcc2"):
    c = [r for r in rows("gpurun_out/%s/**/*counter_collection.csv" % d) if "rc_chain_kernel" in r["Kernel_Name"]]
    if not c:
        continue
    ids = sorted({int(r["Dispatch_Id"]) for r in c})
    pos = {d_: i for i, d_ in enumerate(ids)}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in c:
        i = pos[int(r["Dispatch_Id"])]
        if i >= len(SIZES):
            acc[SIZES[i % len(SIZES)]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for s in SIZES:
        print(" %2dx%-2d " % (s, s) + "  ".join("%s %.3g" % (k.replace("SQ_", ""), sum(v) / len(v)) for k, v in sorted(acc[s].items())))
PY
cat gpurun_out/rcc_run.log | grep -v amdgpu.ids
