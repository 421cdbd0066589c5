#!/bin/bash
cd $GRAFT_REPO_ROOT
for B in 384 512 768 1024; do for M in 1 0; do
  MPCQP_BCR=$M timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-path --no-refactor-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['config']['batch_per_gpu'], 'BCR=$M', round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel'])"
done; done
