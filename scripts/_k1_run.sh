cd $GRAFT_REPO_ROOT
timeout 1700 python scripts/fuzz_share.py 150 500 2>&1 | grep -v amdgpu.ids | grep -v "^ok" | tail -8
timeout 900 python scripts/fuzz_parity.py 71000 800 2>&1 | grep -v amdgpu.ids | grep -v "^ok" | tail -5
timeout 900 python scripts/fuzz_loop.py 71000 400 2>&1 | grep -v amdgpu.ids | grep -v "^ok" | tail -5
