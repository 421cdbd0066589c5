cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_share_factor.py tests/test_gpu_parity.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -4
