cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/g64_stats
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/g64_stats -o s -- python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile --steps 1 --warmup 1 > /dev/null 2>&1
for k in updateTileId trsmPanelId elimGatherTiny chainStep potrfPanelId; do echo "== $k"; python profiles/kdispatch.py gpurun_out/g64_stats/s_results.db $k 26; done > gpurun_out/g64_dispatch.txt
rm -rf gpurun_out/g64_stats
