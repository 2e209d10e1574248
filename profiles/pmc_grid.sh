# FETCH_SIZE / WRITE_SIZE passes of the batched C4 workload (64 x GRID 82x82)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --workload grid82 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
rm -rf gpurun_out/g64_pmc_d gpurun_out/g64_pmc_e
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/g64_pmc_d -o d -- $CMD > gpurun_out/g64_pmc_d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/g64_pmc_e -o e -- $CMD > gpurun_out/g64_pmc_e.log 2>&1
python profiles/summarize_pmc.py gpurun_out/g64_pmc_d/d_results.db gpurun_out/g64_pmc_e/e_results.db > gpurun_out/g64_pmc.txt 2>&1
