# cost of an in-launch dependency between workgroups against a kernel boundary (tools/flag_hop_probe.hip)
cd $GRAFT_REPO_ROOT
timeout 120 tools/flag_hop_probe > gpurun_out/r4_flag_hop.txt 2>&1
cat gpurun_out/r4_flag_hop.txt
