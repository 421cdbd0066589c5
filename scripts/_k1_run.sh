cd $GRAFT_REPO_ROOT
MPCQP_LIB=build_abl/timing.so python scripts/diag_small.py cart_pole 200 2>&1 | grep -v amdgpu.ids | tail -3
