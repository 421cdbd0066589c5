#!/bin/bash
# usage (GPU box): scripts/diag/ab_cfg5.sh <lib.so> [<lib.so> ...]   -- cfg-5 bench of several builds of the library side by side
for b in 256 512 1024; do for lib in "$@"; do python scripts/bench_with_lib.py $lib --workload cfg5 --no-cpu --batch $b > /tmp/o.json; python - <<PY
import json
d=json.loads(open("/tmp/o.json").read().strip().split("\n")[-1])
print("$lib", $b, "loop %.0f stepwise %.0f parity %.0f  iters %.2f/%.2f  frac %.2f" % (d["value"], d["other_path"]["value"], d["parity_setting"]["value"], d["mean_admm_iters"], d["parity_setting"]["mean_admm_iters"], d["roofline"]["frac"]))
PY
done; done
