cd $GRAFT_REPO_ROOT
timeout 1500 python scripts/fuzz_share.py 0 150 2>&1 | grep -v amdgpu.ids | tail -12
