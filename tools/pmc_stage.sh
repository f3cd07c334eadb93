# usage: bash tools/pmc_stage.sh <stage-substring>   (PMC passes for the kernels of one workload stage -> gpurun_out/pmc1, pmc2)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc*
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc1 -o p -- python3 tools/run_stage.py --only $1 --reps 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc2 -o p -- python3 tools/run_stage.py --only $1 --reps 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc1/p_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s n=%4d avg %8.1f us" % (k, len(v), sum(v) / len(v)))
PY
