# usage (GPU box): bash tools/d9_pmc.sh -- memory-side counters of the 9 x 9 search kernel (staging only: VVCGPU_D9_DEBUG=2)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VVCGPU_D9_DEBUG=2
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" ; do
  i=$((i+1)); rm -rf gpurun_out/pmc_d9_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_d9_$i -o p -- python3 tools/run_stage.py --only 9x9 --reps 2 > gpurun_out/pmc_d9_$i.log 2>&1 || tail -3 gpurun_out/pmc_d9_$i.log
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_d9_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sad_dense9" in r["Kernel_Name"]:
            acc[(r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("grid %s wg %s" % k)
    for c, v in sorted(d.items()):
        print("   %-36s %14.0f" % (c, sum(v) / len(v)))
PY
