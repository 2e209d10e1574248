# per-launch sequence of one BAL-871 factor() (rocprofv3 kernel trace): profiles/kseq.py
# usage: bash profiles/r4_trace.sh TAG [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/trace_$TAG
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/trace_$TAG -o s -- python bench.py --no-extras --no-cpu-baseline --no-profile --steps 3 --warmup 1 > gpurun_out/trace_$TAG.json 2> gpurun_out/trace_$TAG.err
DB=$(find gpurun_out/trace_$TAG -name '*.db' | head -1)
python profiles/kseq.py $DB elimFactor > gpurun_out/r4_kseq_$TAG.txt 2>&1
python profiles/stream_timeline.py gpurun_out/trace_$TAG 2 > gpurun_out/r4_timeline_$TAG.txt 2>&1
rm -rf gpurun_out/trace_$TAG
tail -12 gpurun_out/r4_kseq_$TAG.txt
