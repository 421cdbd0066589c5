#!/bin/bash
# Build libmpcqp_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value "$@" \
    -o "${MPCQP_OUT:-../libmpcqp_hip.so}" mpcqp.hip
