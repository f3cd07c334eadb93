# usage (GPU box): bash tools/tz_pmc.sh -- SQ counters of the TZ search kernels (split form): instruction mix and issue / wait cycles per launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_tz1 gpurun_out/pmc_tz2
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_tz1 -o p -- python3 tools/tz_time.py --split --cpu-sample 0 --reps 2 > gpurun_out/pmc_tz1.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_tz2 -o p -- python3 tools/tz_time.py --split --cpu-sample 0 --reps 2 > gpurun_out/pmc_tz2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_tz*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("::")[-1][:30]
        if "tz_" not in n: continue
        acc[(n, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_tz.txt", "w") as o:
    for k, c in sorted(acc.items()):
        line = "%s grid %s lds %s launches %d: " % (k[0], k[1], k[2], len(next(iter(c.values())))) + "  ".join("%s %.3g" % (m, sum(v) / len(v)) for m, v in sorted(c.items()))
        print(line); o.write(line + "\n")
PY
