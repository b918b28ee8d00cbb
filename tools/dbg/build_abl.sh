#!/bin/bash
# Library variants of the point/MLP kernel from -D knobs only (the in-tree generated core is used as it is, nothing in the tree is
# rewritten, so several can be built in parallel):   tools/dbg/build_abl.sh name -DDINER_ABL=3 [-D...]
# Another translation unit: SRC=train tools/dbg/build_abl.sh name -DDINER_DW_INTERLEAVE=1
# -> tools/dbg/libdiner_hip_<name>.so (git-ignored; ships to the GPU box with the snapshot).  A/B them with tools/dbg/ab_bench.py.
set -e
cd "$(dirname "$0")/../../diner_amd/csrc"
name=$1; shift
SRC=${SRC:-points_mlp_f16}
BASE=${BASE:-$SRC}     # the translation unit $SRC stands in for (a patched temporary copy: tools/dbg/build_flat_repro.sh)
UNROLL=""
if [ "$BASE" = points_mlp_f16 ]; then UNROLL=-fno-unroll-loops; fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function -Wno-inline-asm $UNROLL "$@" \
    -c $SRC.hip -o /tmp/${SRC}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libdiner_hip_$name.so $(ls build/*.o | grep -v "/$BASE.o") /tmp/${SRC}_$name.o
echo built tools/dbg/libdiner_hip_$name.so
