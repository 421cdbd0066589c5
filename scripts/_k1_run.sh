cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_share_factor.py -q 2>&1 | grep -v "amdgpu.ids" | tail -5
bash scripts/shared_factor_fetch.sh 4096
