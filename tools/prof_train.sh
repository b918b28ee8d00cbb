#!/bin/bash
# rocprofv3 passes for the training step (tools/bench_train.py); outputs under gpurun_out/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_kt -- python3 $R/tools/bench_train.py > $O/pt_kt.log 2>&1; echo "kernel-trace done"
# (FETCH_SIZE and WRITE_SIZE do not fit one pass -- MI355X_MICROARCH.md, PMC slots: the combined pass stalled in round 3 -- so: two passes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pt_fetch -- python3 $R/tools/bench_train.py > $O/pt_fetch.log 2>&1; echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pt_write -- python3 $R/tools/bench_train.py > $O/pt_write.log 2>&1; echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $O/pt_mfma -- python3 $R/tools/bench_train.py > $O/pt_mfma.log 2>&1
tail -1 $O/pt_kt.log
