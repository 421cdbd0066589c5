#!/bin/bash
# Build libmpcqp_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Two translation units: mpcqp.hip (host side, C ABI, every kernel with 256-thread workgroups) and mpcqp_w8.hip (the latency backend's
# solve kernel with 512-thread workgroups: the same device headers at NT = 512).
# The compiler's per-kernel resource remarks (registers, occupancy, spills) are kept next to the library
# (libmpcqp_hip.kernel_resources.txt; tests/test_kernel_resources.py checks the occupancy the design relies on).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="${MPCQP_OUT:-../libmpcqp_hip.so}"
LOG="${OUT%.so}.build.log"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Rpass-analysis=kernel-resource-usage"
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
: > "$LOG"
( $HIPCC $FLAGS "$@" -c mpcqp_w8.hip -o "$OBJ/w8.o" 2> "$OBJ/w8.log" ) &
W8=$!
$HIPCC $FLAGS "$@" -c mpcqp.hip -o "$OBJ/main.o" 2> "$OBJ/main.log" || { cat "$OBJ/main.log" >&2; exit 1; }
wait $W8 || { cat "$OBJ/w8.log" >&2; exit 1; }
cat "$OBJ/main.log" "$OBJ/w8.log" > "$LOG"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/main.o" "$OBJ/w8.o" 2>> "$LOG" || { cat "$LOG" >&2; exit 1; }
grep -E "error|warning" "$LOG" | grep -v "Rpass-analysis" >&2 || true
grep "Rpass-analysis=kernel-resource-usage" "$LOG" | sed 's/.*remark: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' > "${OUT%.so}.kernel_resources.txt"
rm -f "$LOG"
