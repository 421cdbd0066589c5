cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_examples.py -q 2>&1 | grep -v amdgpu.ids | tail -5
