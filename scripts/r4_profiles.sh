#!/bin/bash
# round-4 evidence in one GPU call: rocprofv3 kernel stats + HBM counters (separate passes) of the driver's bench command for cfg-3,
# for the HBM-only leg's shape (cfg-3 at batch 4096, key cfg3_b4096) and for cfg-5; the same for the latency backend (256 instances);
# the bench lines.  Summaries land in gpurun_out/r4*; the ones to be judged are copied into profiles/.
cd $GRAFT_REPO_ROOT; O=gpurun_out
bash scripts/profile_round.sh r4 cfg3 > $O/r4_profile_cfg3.log 2>&1
bash scripts/profile_round.sh r4 cfg5 > $O/r4_profile_cfg5.log 2>&1
bash scripts/profile_round.sh r4b4096 cfg3 --batch 4096 > $O/r4_profile_b4096.log 2>&1
bash scripts/profile_round.sh r4b256 cfg3 --batch 256 > $O/r4_profile_b256.log 2>&1
python - <<PY
import json
o = json.load(open('$O/r4_pmc_hbm_traffic.json'))
o['cfg3_b4096'] = json.load(open('$O/r4b4096_pmc_hbm_traffic.json'))['cfg3']
o['cfg3_b256'] = json.load(open('$O/r4b256_pmc_hbm_traffic.json'))['cfg3']
json.dump(o, open('$O/r4_pmc_hbm_traffic_all.json', 'w'), indent=1)
PY
tail -4 $O/r4_profile_cfg3.log $O/r4_profile_b4096.log $O/r4_profile_cfg5.log
