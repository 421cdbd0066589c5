cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gaps.py -m gpu -q -x -s -k "north_star" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-600
