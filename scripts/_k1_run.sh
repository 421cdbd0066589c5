cd $GRAFT_REPO_ROOT
MPCQP_BENCH_FORCE_PG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-path --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r6_bench_rccl_one_rank.json
MPCQP_BENCH_FORCE_PG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-path --no-cpu-baseline --path stepwise 2>/dev/null | tail -n 1 > gpurun_out/r6_bench_rccl_one_rank_stepwise.json
MPCQP_BENCH_FORCE_PG=1 timeout 600 python bench.py --gpus 1 --steps 40 --warmup 20 --shared-model --backend sweeps --batch 4096 2>/dev/null | tail -n 1 > gpurun_out/r6_bench_rccl_one_rank_shared_model.json
python - <<'P'
import json
for f in ('r6_bench_rccl_one_rank', 'r6_bench_rccl_one_rank_stepwise', 'r6_bench_rccl_one_rank_shared_model'):
    d = json.load(open('gpurun_out/%s.json' % f)); print(f, d['value'], d['collective_backend'], d['ranks_seen'], d['per_rank'])
P
