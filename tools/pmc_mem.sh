# usage (GPU box): bash tools/pmc_mem.sh r02 -- memory-side counters (L1 -> L2 requests, L2 hits / misses, requests to the fabric, L1 pending stalls) for
# every kernel of the canonical workload -> profiles/<tag>_pmc_mem.csv (one row per kernel and grid, per-launch averages)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_mem1 gpurun_out/pmc_mem2
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d gpurun_out/pmc_mem1 -o p -- python3 tools/run_stage.py --reps 2 > gpurun_out/pmc_mem1.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_mem2 -o p -- python3 tools/run_stage.py --reps 2 > gpurun_out/pmc_mem2.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys, os
tag = sys.argv[1]
sys.path.insert(0, "profiles")
from summarize import short
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_mem*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_mem1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
        dur[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = sorted({n for d in acc.values() for n in d})
keys = sorted(acc, key=lambda k: -sum(dur.get(k, [0])))
out = os.path.join("profiles", tag + "_pmc_mem.csv")
with open(out, "w") as fh:
    fh.write("kernel,grid_threads,avg_us," + ",".join(names) + ",l2_miss_frac,l1_to_l2_GBps_at_128B\n")
    for k in keys:
        d = dur.get(k) or [0]
        us = sum(d) / len(d) / 1e3
        v = {n: (sum(acc[k][n]) / len(acc[k][n]) if acc[k][n] else 0.0) for n in names}
        hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        rd, wr = v.get("TCP_TCC_READ_REQ_sum", 0.0), v.get("TCP_TCC_WRITE_REQ_sum", 0.0)
        fh.write('"%s",%d,%.2f,%s,%.3f,%.0f\n' % (k[0], k[1], us, ",".join("%.0f" % v[n] for n in names), miss / (hit + miss) if hit + miss else 0.0,
                                                   (rd + wr) * 128 / (us * 1e-6) / 1e9 if us else 0.0))
print(open(out).read()[:6000])
PY
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_pmc_mem.csv gpurun_out/profiles_out/
