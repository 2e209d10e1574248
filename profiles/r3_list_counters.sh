cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/r3_counters_raw.txt 2>&1 || rocprofv3 --list-avail > gpurun_out/r3_counters_raw.txt 2>&1
wc -l gpurun_out/r3_counters_raw.txt; grep -oE "\b(TA|TCP|TCC|MALL)_[A-Za-z0-9_]+" gpurun_out/r3_counters_raw.txt | sort -u | tr '\n' ' ' | head -c 6000
