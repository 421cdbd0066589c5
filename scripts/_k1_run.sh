cd $GRAFT_REPO_ROOT
timeout 120 python scripts/diag/rccl_banner.py 2>/dev/null | cat
