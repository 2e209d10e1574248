# FETCH_SIZE / TCC_EA0_RDREQ calibration for the 216-byte-block gather pattern (tools/fetch_calib.hip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/calib
tools/fetch_calib > gpurun_out/calib/plain.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/calib/a -o a -- tools/fetch_calib > gpurun_out/calib/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/calib/b -o b -- tools/fetch_calib > gpurun_out/calib/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/calib/c -o c -- tools/fetch_calib > gpurun_out/calib/c.log 2>&1
( cat gpurun_out/calib/plain.txt; python profiles/summarize_pmc.py $(find gpurun_out/calib/a gpurun_out/calib/b -name '*.db'); find gpurun_out/calib/c -name '*kernel_stats*' | head -1 | xargs cat ) > gpurun_out/r4_fetch_calib.txt 2>&1
rm -rf gpurun_out/calib/*/*.db
