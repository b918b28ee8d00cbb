#!/bin/bash
# rocprofv3 passes of one bench configuration (run on the GPU box via gpurun); raw output under gpurun_out/prof_<cfg>/,
# summaries are then written by tools/pmc_summary.py into profiles/.   usage: tools/prof.sh [cfg3] [extra bench args]
# Counters are collected in separate passes and never combined with a trace domain (profiles/README.md).
set -e
CFG=${1:-cfg3}; shift || true
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$CFG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config $CFG --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 2 --warmup 1 > $O/kt.log 2>&1; echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1; echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/write.log 2>&1; echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B --steps 1 --warmup 0 > $O/mfma.log 2>&1; echo "mfma done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- $B --steps 1 --warmup 0 > $O/tcc.log 2>&1; echo "tcc done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/sq -- $B --steps 1 --warmup 0 > $O/sq.log 2>&1 || echo "sq pass failed (optional)"
echo "all passes done"
