# End-of-round evidence, all from the default bench workload (run through gpurun from the repo root):
#   kernel-trace + stats of `python bench.py`, the same build's plain bench line, a per-stream
#   timeline, and the PMC passes (one counter group per run, kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01_v5}
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o s -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_tl -o t -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_tl.log 2>&1
python profiles/timeline_streams.py gpurun_out/${TAG}_tl 0 0 > gpurun_out/${TAG}_timeline.txt 2>&1
bash profiles/collect_pmc.sh
ls gpurun_out/${TAG}_stats gpurun_out/pmc_d gpurun_out/pmc_e
