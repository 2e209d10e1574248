# verdict item 2: per-op samples on today's kernels -> fit -> merge sweep per batch size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/model
timeout 900 python tools/op_stats_dump.py gpurun_out/model/r06_opstats 2> gpurun_out/model/dump.log
python tools/fit_computation_model.py gpurun_out/model/r06_opstats | tee gpurun_out/model/r06_model_fit.txt | tail -1 > gpurun_out/model/r06_model_fit.json
head -4 gpurun_out/model/r06_model_fit.txt
timeout 1500 python tools/model_batch_sweep.py gpurun_out/model/r06_model_fit.json grid82 2>&1 | tee gpurun_out/model/r06_batch_sweep_grid82.txt
