#!/bin/bash
# Reproducer of the FLAT problem: the failing build of the hunt = the view-sum slab back on generic pointers
# (hipcc then emits flat_load / flat_store) + 4-wave geometry + DINER_BIAS_C.  The product source carries no knob for it: this script
# compiles a patched temporary copy.   tools/dbg/build_flat_repro.sh [stores|loads|both|stores_nop]   -> tools/dbg/libdiner_hip_flat_<which>.so
# (stores_nop: the flat stores each followed by s_nop 3 -- more wait states before anything can overwrite their data registers)
# then:  REPS=16 python tools/dbg/g1_race.py tools/dbg/libdiner_hip_flat_both.so        (13-15 of 16 processes hit at commit cbb8e8d; 0 of 16 on the round-3
# final tree: the failure also needs the register allocation and timing of that commit -- check it out to reproduce)
set -e
which=${1:-both}
cd "$(dirname "$0")/../../diner_amd/csrc"
tmp=points_mlp_f16_flatrepro
cp points_mlp_f16.hip $tmp.hip
if [ "$which" = loads ] || [ "$which" = both ]; then sed -i 's/return \*(const g_f32x4 \*)p;/return *(const f32x4 *)p;/' $tmp.hip; fi
if [ "$which" = stores ] || [ "$which" = both ]; then sed -i 's/^    asm volatile("global_store_dwordx4 %0, %1, off.*$/    *(f32x4 *)p = v;/' $tmp.hip; fi
if [ "$which" = stores_nop ]; then sed -i 's/^    asm volatile("global_store_dwordx4 %0, %1, off.*$/    *(f32x4 *)p = v; asm volatile("s_nop 3");/' $tmp.hip; fi
if cmp -s $tmp.hip points_mlp_f16.hip; then echo "patch did not apply"; rm -f $tmp.hip; exit 1; fi
trap 'rm -f '"$PWD/$tmp.hip" EXIT
SRC=$tmp BASE=points_mlp_f16 ../../tools/dbg/build_abl.sh flat_$which -DDINER_GEOM_WAVES=4 -DDINER_BIAS_C=1
