cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "three_quarter or persistent_queue" 2>&1 | tail -5
