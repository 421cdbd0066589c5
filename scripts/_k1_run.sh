cd $GRAFT_REPO_ROOT
python scripts/latency_single.py 2>&1 | grep -v amdgpu.ids | tail -3
python scripts/latency_single.py 2>&1 | grep -v amdgpu.ids | tail -3 | head -1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
