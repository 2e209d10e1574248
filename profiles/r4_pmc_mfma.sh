# MFMA instruction counts per kernel of one BAL-871 factor (rocprofv3 --pmc, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA -d gpurun_out/pmc_g -o g -- $CMD > gpurun_out/pmc_g.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_g/g_results.db > gpurun_out/r04_pmc_mfma.txt 2>&1
rm -rf gpurun_out/pmc_g
cat gpurun_out/r04_pmc_mfma.txt | head -20
