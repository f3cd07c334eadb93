# usage (GPU box): bash tools/pmc_one.sh <launch-group substring> <tag>   -- SQ counter passes for the kernels of one launch group -> gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -o p -- python3 tools/run_stage.py --only $1 --reps 2 > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o p -- python3 tools/run_stage.py --only $1 --reps 2 > /dev/null 2>&1 &&
python3 profiles/summarize_pmc.py gpurun_out zz_$2 > gpurun_out/pmc_$2.txt 2>&1; rm -f profiles/zz_$2_pmc_sq.csv
