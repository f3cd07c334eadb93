# usage (GPU box): bash tools/r5q_ab.sh -- A/B timing of the 39 x 39 raster searches: quad form (sad_raster5q_kernel) against the pair form (VVCGPU_NO_R5Q=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in q:0; do
  unset VVCGPU_NO_R5Q VVCGPU_R5Q_SPLIT VVCGPU_R5C_RPS VVCGPU_NO_R5GQ VVCGPU_R5Q_NOTOUCH
  case $v in old:*) export VVCGPU_NO_R5Q=1;; rps15:*) export VVCGPU_R5C_RPS=15;; g:*) export VVCGPU_NO_R5GQ=1;; nt:*) export VVCGPU_R5Q_NOTOUCH=1;; esac
  t=${v#*:}; if [ "$t" != "0" ]; then export VVCGPU_R5Q_SPLIT=$t; fi
  rm -rf gpurun_out/prof_r5q
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r5q -o q -- python3 tools/run_stage.py --only 39x39 --reps 6 > gpurun_out/r5q_run.log 2>&1
  echo "variant $v"
  python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_r5q/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sad_raster5" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0].split("::")[-1][:40], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[len(v) // 3:]
    print("   %-40s grid %8s wg %4s lds %6s : avg %7.1f us  min %7.1f" % (k[0], k[1], k[2], k[3], sum(v) / len(v), min(v)))
PY
done
