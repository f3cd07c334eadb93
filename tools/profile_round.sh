# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02
# kernel trace + stats, FETCH_SIZE and WRITE_SIZE PMC passes (separate runs, pmc never combined with other trace domains), the fabric read /
# write request counters BY REQUEST SIZE (TCC_EA0_RDREQ_32B / _64B / _128B: the exact bytes behind FETCH_SIZE), SQ
# instruction / busy counters for every kernel of the workload, then the bench line; output under gpurun_out/ (summaries are
# written into profiles/ by profiles/summarize.py and profiles/summarize_pmc.py and copied back through gpurun_out/profiles_out)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_* gpurun_out/pmc_sq*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o $TAG -- python3 bench.py --steps 3 --warmup 1 --pictures-per-step 4 --no-cpu-baseline --no-real-mix --no-input-stream --no-depquant-leg --serial > gpurun_out/prof_kt.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -o $TAG -- python3 bench.py --steps 1 --warmup 0 --pictures-per-step 3 --no-cpu-baseline --no-real-mix --no-input-stream --no-depquant-leg --serial > gpurun_out/prof_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -o $TAG -- python3 bench.py --steps 1 --warmup 0 --pictures-per-step 3 --no-cpu-baseline --no-real-mix --no-input-stream --no-depquant-leg --serial > gpurun_out/prof_write.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d gpurun_out/prof_rdreq -o $TAG -- python3 bench.py --steps 1 --warmup 0 --pictures-per-step 3 --no-cpu-baseline --no-real-mix --no-input-stream --no-depquant-leg --serial > gpurun_out/prof_rdreq.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d gpurun_out/prof_wrreq -o $TAG -- python3 bench.py --steps 1 --warmup 0 --pictures-per-step 3 --no-cpu-baseline --no-real-mix --no-input-stream --no-depquant-leg --serial > gpurun_out/prof_wrreq.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -o p -- python3 tools/run_stage.py --reps 2 > gpurun_out/pmc_sq1.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o p -- python3 tools/run_stage.py --reps 2 > gpurun_out/pmc_sq2.log 2>&1 &&
python3 profiles/summarize.py gpurun_out $TAG > gpurun_out/summarize.log 2>&1 &&
python3 profiles/summarize_pmc.py gpurun_out $TAG >> gpurun_out/summarize.log 2>&1 &&
python3 tools/mfma_vs_dot2.py > profiles/${TAG}_mfma_vs_dot2.txt 2> gpurun_out/mfma_vs_dot2.err &&
python3 tools/shape_mix_time.py 8388608 2> gpurun_out/shape_mix.err | grep -v amdgpu.ids > profiles/${TAG}_shape_mix.txt &&
python3 bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/bench.err
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/ 2>/dev/null
cp gpurun_out/${TAG}_bench_n1.json gpurun_out/profiles_out/ 2>/dev/null
tail -c 600 gpurun_out/${TAG}_bench_n1.json
