#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/prof_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/prof_sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/prof_tcc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_tcc.log 2>&1
echo done
