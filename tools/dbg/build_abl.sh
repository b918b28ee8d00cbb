#!/bin/bash
# Library variants of the point/MLP kernel from -D knobs only (the in-tree generated core is used as it is, nothing in the tree is
# rewritten, so several can be built in parallel):   tools/dbg/build_abl.sh name -DDINER_ABL=3 [-D...]
# -> tools/dbg/libdiner_hip_<name>.so (git-ignored; ships to the GPU box with the snapshot).  A/B them with tools/dbg/ab_bench.py.
set -e
cd "$(dirname "$0")/../../diner_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function -Wno-inline-asm -fno-unroll-loops "$@" \
    -c points_mlp_f16.hip -o /tmp/f16_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libdiner_hip_$name.so $(ls build/*.o | grep -v points_mlp_f16.o) /tmp/f16_$name.o
echo built tools/dbg/libdiner_hip_$name.so
