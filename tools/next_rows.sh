# usage (GPU box, repo root): bash tools/next_rows.sh r01   -- timing lines of every "next" row (N1-N4) + their rocprofv3 kernel stats
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_next
{ python3 tools/n13_time.py; python3 tools/tz_time.py --cpu-sample 600; python3 tools/tz_time.py --split --cpu-sample 0; python3 tools/tz_time.py --team 1 --cpu-sample 0; python3 tools/n4_time.py; python3 tools/intra_search_time.py | head -2; } 2>/dev/null > gpurun_out/${TAG}_next_rows.txt
cat gpurun_out/${TAG}_next_rows.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_next -o $TAG -- python3 tools/tz_time.py --cpu-sample 0 --reps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_next4 -o $TAG -- python3 tools/n4_time.py > /dev/null 2>&1
find gpurun_out/prof_next gpurun_out/prof_next4 -name "*kernel_stats.csv" | head
