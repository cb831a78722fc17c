#!/bin/bash
# SQ counters of the hot-path kernels (one --pmc pass, kernel-trace only): instruction counts and wave-cycle buckets.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-views 0 --loop-views 0 --views 32"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d ${OUT} -o s -- $CMD > ${OUT}.log 2>&1
ls ${OUT}
