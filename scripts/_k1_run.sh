cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_controller_map.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout 300 python examples/controller_map_monte_carlo.py 2>&1 | grep -v amdgpu.ids
timeout 300 python examples/controller_map_monte_carlo.py --eps 1e-9 2>&1 | grep -v amdgpu.ids
