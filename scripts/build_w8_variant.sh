#!/bin/bash
# Development: a variant of the 512-thread translation unit (mpcqp_w8.hip) linked against a cached object of mpcqp.hip, ~15 s instead of 2 min.
#   scripts/build_w8_variant.sh <out.so> [flags for BOTH units the first time, e.g. -DMPCQP_RUN_TIMING] -- [flags for mpcqp_w8.hip only]
# The cached main object lives in build_abl/obj/main_<hash of the common flags>.o
set -e
cd "$(dirname "$0")/../pympc_amd/csrc"
OUT=$(realpath -m "../../$1"); shift
COMMON=(); W8=(); seen=0
for a in "$@"; do if [ "$a" == "--" ]; then seen=1; elif [ $seen == 0 ]; then COMMON+=("$a"); else W8+=("$a"); fi; done
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
KEY=$(echo "${COMMON[@]}" | md5sum | cut -c1-8)
MAIN=../../build_abl/obj/main_$KEY.o
if [ ! -f "$MAIN" ] || [ mpcqp.hip -nt "$MAIN" ] || [ -n "$(find . -name '*.h' -newer "$MAIN" ! -name mpcqp_latw.h)" ]; then
    /opt/rocm/bin/hipcc $FLAGS "${COMMON[@]}" -c mpcqp.hip -o "$MAIN"
fi
/opt/rocm/bin/hipcc $FLAGS "${COMMON[@]}" "${W8[@]}" -c mpcqp_w8.hip -o /tmp/w8_variant_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$MAIN" /tmp/w8_variant_$$.o
rm -f /tmp/w8_variant_$$.o
