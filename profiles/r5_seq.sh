# launch sequence (rocprofv3 kernel trace) of one factor(): bash profiles/r5_seq.sh TAG "bench args" [ENV=..]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ARGS=$2; shift; shift
rm -rf gpurun_out/${TAG}_stats
env "$@" timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_stats -o s -- python bench.py $ARGS --no-extras --no-cpu-baseline --no-profile --steps 2 --warmup 1 > gpurun_out/${TAG}.log 2>&1
python profiles/kseq.py gpurun_out/${TAG}_stats/s_results.db elimFactor > gpurun_out/${TAG}_seq.txt 2>&1
rm -rf gpurun_out/${TAG}_stats
tail -1 gpurun_out/${TAG}.log | cut -c1-200
