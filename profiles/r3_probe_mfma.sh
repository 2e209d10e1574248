cd $GRAFT_REPO_ROOT
timeout 120 tools/mfma_sustained_probe
