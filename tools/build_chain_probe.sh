#!/bin/bash
# builds tools/chain_probe: the generated GEMM cores in three variants + the probe
set -e
cd "$(dirname "$0")/.."
G=diner_amd/csrc/gen_f16_core.py
{ python3 $G 4 --ns=d4; python3 $G 2 --ns=d2; python3 $G 4 --ns=d4nl --noload; } > tools/chain_probe_cores.inc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-inline-asm -o tools/chain_probe tools/chain_probe.hip
