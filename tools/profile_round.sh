# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
# kernel trace + stats, FETCH_SIZE and WRITE_SIZE PMC passes (separate runs), the bench line; output under gpurun_out/
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o $TAG -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial > gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -o $TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -o $TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial > gpurun_out/prof_write.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/bench.err
tail -c 400 gpurun_out/${TAG}_bench_n1.json
