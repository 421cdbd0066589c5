cd $GRAFT_REPO_ROOT
timeout 300 python examples/closed_loop_kalman.py --eps 1e-10 --device-loop 32 2>&1 | grep -v amdgpu.ids | tail
timeout 300 python examples/closed_loop_kalman.py --device-loop 256 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_examples.py -q 2>&1 | grep -v amdgpu.ids | tail -5
