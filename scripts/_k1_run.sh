cd $GRAFT_REPO_ROOT
MPCQP_BENCH_FORCE_PG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-path --no-cpu-baseline > gpurun_out/fpg.out 2> gpurun_out/fpg.err; echo rc=$?
echo "last line starts with: $(tail -n 1 gpurun_out/fpg.out | cut -c1-60)"; echo "lines: $(wc -l < gpurun_out/fpg.out)"; head -n 3 gpurun_out/fpg.out | cut -c1-80
timeout 900 python -m pytest tests/test_gpu_rccl_single_rank.py -q 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/plain.out 2>/dev/null; echo "plain last: $(tail -n 1 gpurun_out/plain.out | cut -c1-40) lines $(wc -l < gpurun_out/plain.out)"
