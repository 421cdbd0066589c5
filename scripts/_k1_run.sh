cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rccl_single_rank.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout 300 python bench.py --shared-model --backend sweeps --batch 4096 --steps 40 --warmup 20 | tail -n 1 | cut -c1-1500
