#!/bin/bash
# HBM traffic of the render kernels per launch, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only, no other tracing domains.
# FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads -> reported raw and x2.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/hbm_$1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-views 0 --loop-views 0 --views 32"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ${OUT}_fetch -o f -- $CMD > ${OUT}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d ${OUT}_write -o w -- $CMD > ${OUT}_write.log 2>&1
ls ${OUT}_fetch ${OUT}_write
