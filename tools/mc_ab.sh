# usage (GPU box): bash tools/mc_ab.sh -- kernel times of the MC launches of the canonical workload in isolation (luma, Cb, Cr; fast + generic kernel)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in now; do
  rm -rf gpurun_out/prof_mc
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mc -o t -- python3 tools/run_stage.py --only mc/ --reps 6 > gpurun_out/mc_run.log 2>&1
  echo "variant $v"
  python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_mc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][-18:], r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[len(v) // 3:]
    print("   %-20s grid %9s calls %3d avg %7.1f us min %7.1f  [%s]" % (k[0], k[1], len(v), sum(v) / len(v), min(v), " ".join("%.0f" % x for x in v[:6])))
PY
done
