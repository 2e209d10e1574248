cd $GRAFT_REPO_ROOT
timeout 300 tools/bulk_pipeline_probe q
