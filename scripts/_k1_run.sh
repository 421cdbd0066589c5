cd $GRAFT_REPO_ROOT
timeout 900 python scripts/fuzz_parity.py 70000 1500 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python scripts/fuzz_loop.py 70000 600 2>&1 | grep -v amdgpu.ids | tail -2
