#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box via gpurun); outputs under gpurun_out/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_kt.log 2>&1
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_mfma.log 2>&1
echo "mfma done"
find $O -name "*.csv" | head -40
