#!/bin/bash
# Development: variant builds from cached objects, one per translation unit and flag set (no staleness logic: delete build_abl/obj/<name>.o to rebuild).
#   scripts/build_variant_objs.sh main <tag> [flags]      -> build_abl/obj/M_<tag>.o (mpcqp.hip, ~2 min)
#   scripts/build_variant_objs.sh w8 <tag> [flags]        -> build_abl/obj/W_<tag>.o (mpcqp_w8.hip, ~30 s)
#   scripts/build_variant_objs.sh link <name> <main tag> <w8 tag>   -> build_abl/<name>.so  (run with scripts/with_lib.py, lat_phase.py, lat_ab.py --lib)
cd /root/repo/pympc_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
O=/root/repo/build_abl/obj
main() { tag=$1; shift; /opt/rocm/bin/hipcc $FLAGS "$@" -c mpcqp.hip -o $O/M_$tag.o 2> $O/M_$tag.log || echo "FAILED M_$tag"; }
w8() { tag=$1; shift; /opt/rocm/bin/hipcc $FLAGS "$@" -c mpcqp_w8.hip -o $O/W_$tag.o 2> $O/W_$tag.log || echo "FAILED W_$tag"; }
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_abl/$1.so $O/M_$2.o $O/W_$3.o || echo "FAILED link $1"; }
"$@"
