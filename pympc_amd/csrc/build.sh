#!/bin/bash
# Build libmpcqp_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# The compiler's per-kernel resource remarks (registers, occupancy, spills) are kept next to the library
# (libmpcqp_hip.kernel_resources.txt; tests/test_kernel_resources.py checks the occupancy the design relies on).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="${MPCQP_OUT:-../libmpcqp_hip.so}"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Rpass-analysis=kernel-resource-usage "$@" \
    -o "$OUT" mpcqp.hip 2> "${OUT%.so}.build.log" || { cat "${OUT%.so}.build.log" >&2; exit 1; }
grep -E "error|warning" "${OUT%.so}.build.log" | grep -v "Rpass-analysis" >&2 || true
grep "Rpass-analysis=kernel-resource-usage" "${OUT%.so}.build.log" | sed 's/.*remark: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' > "${OUT%.so}.kernel_resources.txt"
rm -f "${OUT%.so}.build.log"
