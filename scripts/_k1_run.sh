cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cost_invariant.py -q 2>&1 | grep -v amdgpu.ids | tail -8
