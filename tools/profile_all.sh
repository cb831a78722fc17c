#!/bin/bash
# One GPU-box call -> everything profiles/ needs for ONE workload of bench.py, all from the same build:
#   [WORKLOAD="P W H sh_degree"] tools/profile_all.sh TAG [bench.py workload flags, e.g. --sh-degree 3]
#   1. kernel trace (rocprofv3 --kernel-trace --stats) of `bench.py --steps 3 --warmup 1`   -> gpurun_out/TAG_trace
#   2. SQ instruction / wave-cycle counters (one --pmc pass, kernel-trace only)              -> gpurun_out/TAG_sq
#   3. FETCH_SIZE and 4. WRITE_SIZE in SEPARATE --pmc passes (TCC slots; MI355X_MICROARCH.md, HBM section)
# No other tracing domain is ever combined with --pmc.  The build id of the library is recorded beside the results;
# tools/profile_summary.py turns the .db files into the tables under profiles/.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $R/gpurun_out
BENCH="python $R/bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 $*"
python -c "import sys; sys.path.insert(0, '$R/gaussian-garments_amd'); from ggsplat import _lib; print(_lib.build_id())" > ${OUT}_build_id.txt
echo "$*" > ${OUT}_args.txt
timeout 600 rocprofv3 --kernel-trace --stats -d ${OUT}_trace -o t -- $BENCH --steps 3 --warmup 1 > ${OUT}_trace.log 2>&1 || echo "trace pass failed" >> ${OUT}_trace.log
if [ -n "$ONLY_TRACE" ]; then
  grep -h '^{' ${OUT}_trace.log | tail -1 > ${OUT}_bench.json || true
  python $R/tools/profile_summary.py $TAG $R/gpurun_out/prof_${TAG} $WORKLOAD || true
  rm -rf ${OUT}_trace
  exit 0
fi
PMC="$BENCH --steps 1 --warmup 0 --views ${PMC_VIEWS:-32}"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d ${OUT}_sq -o s -- $PMC > ${OUT}_sq.log 2>&1 || echo "sq pass failed" >> ${OUT}_sq.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ${OUT}_fetch -o f -- $PMC > ${OUT}_fetch.log 2>&1 || echo "fetch pass failed" >> ${OUT}_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d ${OUT}_write -o w -- $PMC > ${OUT}_write.log 2>&1 || echo "write pass failed" >> ${OUT}_write.log
grep -h '^{' ${OUT}_trace.log | tail -1 > ${OUT}_bench.json || true
# summarise on the box and drop the databases: the merge back into gpurun_out/ is capped at 64 MiB
python $R/tools/profile_summary.py $TAG $R/gpurun_out/prof_${TAG} $WORKLOAD || true
rm -rf ${OUT}_trace ${OUT}_sq ${OUT}_fetch ${OUT}_write
for f in ${OUT}_trace.log ${OUT}_sq.log ${OUT}_fetch.log ${OUT}_write.log; do tail -5 $f > $f.tail; rm -f $f; done
ls -la $R/gpurun_out | grep ${TAG} | head -20
