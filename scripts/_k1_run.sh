cd $GRAFT_REPO_ROOT
EXTRA="--backend sweeps --shared-model" bash scripts/pmc_sq.sh r6shared4096 4096 | tail -12
EXTRA="--backend sweeps" bash scripts/pmc_sq.sh r6sw4096 4096 | tail -12
