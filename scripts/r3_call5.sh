#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 300 python bench.py --workload cfg2 2>/dev/null
timeout 300 python bench.py --workload notebook 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
