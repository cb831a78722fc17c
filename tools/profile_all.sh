#!/bin/bash
# One GPU-box call -> everything profiles/ needs for ONE workload of bench.py, all from the same build:
#   [WORKLOAD="P W H sh_degree"] [PASSES="trace sq valu fetch write"] tools/profile_all.sh TAG [bench.py workload flags, e.g. --sh-degree 3]
#   trace  kernel trace (rocprofv3 --kernel-trace --stats) of `bench.py --steps 3 --warmup 1`    -> gpurun_out/TAG_trace
#   sq     SQ instruction / wave-cycle counters (one --pmc pass, kernel-trace only)               -> gpurun_out/TAG_sq
#   lds    LDS pipe: instructions, busy / stall / bank-conflict cycles (one --pmc pass)
#   valu   VALU busy / lane activity: raw counters, then rocprofv3's derived VALUBusy / VALUUtilization (two --pmc passes)
#   atomic, vmem   L2 atomic requests / tag stalls; vector-memory instructions and what the waves wait for (round 6; generic tables)
#   fetch, write   FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots; MI355X_MICROARCH.md, HBM section)
# No other tracing domain is ever combined with --pmc.  The build id of the library is recorded beside the results;
# tools/profile_summary.py turns the .db files into the tables under profiles/ (only for the passes that ran).
# ONLY_TRACE=1 is short for PASSES=trace.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $R/gpurun_out
PASSES=${PASSES:-trace sq valu fetch write}
[ -n "$ONLY_TRACE" ] && PASSES=trace
BENCH="python $R/bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 $*"
# counters are sampled per dispatch over the whole device: launch sets must NOT overlap on two streams while they are collected
PMC="$BENCH --pipeline 0 --steps 1 --warmup 0 --views ${PMC_VIEWS:-32}"
python -c "import sys; sys.path.insert(0, '$R/gaussian-garments_amd'); from ggsplat import _lib; print(_lib.build_id())" > ${OUT}_build_id.txt
echo "$*" > ${OUT}_args.txt
pmc_pass() {   # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d ${OUT}_$name -o $name -- $PMC > ${OUT}_$name.log 2>&1 || echo "$name pass failed" >> ${OUT}_$name.log
}
for p in $PASSES; do
  case $p in
    trace) timeout 600 rocprofv3 --kernel-trace --stats -d ${OUT}_trace -o t -- $BENCH --steps 3 --warmup 1 > ${OUT}_trace.log 2>&1 || echo "trace pass failed" >> ${OUT}_trace.log
           grep -h '^{' ${OUT}_trace.log | tail -1 > ${OUT}_bench.json || true ;;
    sq)    pmc_pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU ;;
    valu)  pmc_pass valu SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
           pmc_pass valud VALUBusy VALUUtilization ;;
    lds)   pmc_pass lds SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES ;;
    atomic) pmc_pass atomic TCC_ATOMIC_sum TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum ;;      # L2 side of the float atomics (round 6)
    vmem)  pmc_pass vmem SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES ;;
    fetch) pmc_pass fetch FETCH_SIZE ;;
    write) pmc_pass write WRITE_SIZE ;;
  esac
done
# summarise on the box and drop the databases: the merge back into gpurun_out/ is capped at 64 MiB
python $R/tools/profile_summary.py $TAG $R/gpurun_out/prof_${TAG} $WORKLOAD || true
rm -rf ${OUT}_trace ${OUT}_sq ${OUT}_lds ${OUT}_valu ${OUT}_valud ${OUT}_fetch ${OUT}_write ${OUT}_atomic ${OUT}_vmem
for f in ${OUT}_*.log; do tail -5 $f > $f.tail; rm -f $f; done
ls -la $R/gpurun_out | grep ${TAG} | head -20
