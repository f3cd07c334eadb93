# usage (GPU box): bash tools/d9_ab.sh  -- A/B timing of the 9 x 9 search: row form (sad_dense9_kernel) with several workgroup sizes against the
# position-per-lane form (VVCGPU_NO_D9=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in d9:0 d9:64 d9:128 nocompute:0 nocompute:64; do
  unset VVCGPU_NO_D9 VVCGPU_D9_T VVCGPU_D9_DEBUG
  case $v in old:*) export VVCGPU_NO_D9=1;; nofill:*) export VVCGPU_D9_DEBUG=1;; nocompute:*) export VVCGPU_D9_DEBUG=2;; esac
  t=${v#*:}; if [ "$t" != "0" ]; then export VVCGPU_D9_T=$t; fi
  rm -rf gpurun_out/prof_d9
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_d9 -o d9 -- python3 tools/run_stage.py --only 9x9 --reps 10 > gpurun_out/d9_run.log 2>&1
  echo "variant $v"
  python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_d9/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sad_dense" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0].split("::")[-1], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[len(v) // 3:]
    print("   %-20s grid %8s wg %4s : avg %7.1f us  min %7.1f" % (k[0], k[1], k[2], sum(v) / len(v), min(v)))
PY
done
