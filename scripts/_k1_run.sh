cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_controller_map.py -m gpu -q -x 2>&1 | tail -5
python examples/controller_map_monte_carlo.py 2>&1 | grep -v amdgpu.ids | tail -4
python examples/controller_map_monte_carlo.py --eps 1e-9 --n 4000 --sequential 200 2>&1 | grep -v amdgpu.ids | tail -4
