#!/bin/bash
# job.sh STEP [STEP ...] -- the GPU-box jobs of a round as ONE parameterised script (replaces the per-experiment r3_*.sh files).
# Run through gpurun from the repo root:  gpurun --timeout 900 -- 'bash tools/dbg/job.sh ubench_lds census ab'
# Every step writes under gpurun_out/ (merged back by gpurun).  Environment: TAG (file-name prefix, default r04),
# TESTS (pytest selection for `partests`), LIBS (variant names for `ab` / `partests`, default: all of csrc/variants/*.so),
# BENCH_ARGS (extra bench.py flags for `ab`).  seltests: TESTS = the selection.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
TAG=${TAG:-r05}
OUT=gpurun_out
mkdir -p $OUT
VAR=gaussian-garments_amd/csrc/variants
libs() { if [ -n "$LIBS" ]; then for l in $LIBS; do echo $VAR/$l.so; done; else ls $VAR/*.so; fi; }
for step in "$@"; do
  echo "=== $step ==="
  case $step in
    ubench_lds)      # LDS float-atomic rates + the transposed-reduction block (tools/ubench/lds_atomics.hip)
      (cd tools/ubench && timeout 300 ./lds_atomics) > $OUT/${TAG}_lds_atomics.txt 2>&1; tail -45 $OUT/${TAG}_lds_atomics.txt ;;
    ubench_systolic) (cd tools/ubench && timeout 300 ./systolic_parts) > $OUT/${TAG}_systolic_parts.txt 2>&1; tail -40 $OUT/${TAG}_systolic_parts.txt ;;
    census)          timeout 600 python tools/dbg/fold_census.py > $OUT/${TAG}_fold_census.txt 2>&1; cat $OUT/${TAG}_fold_census.txt ;;
    partests)        # parity tests against a variant library
      for f in $(libs); do echo "--- $f"; GGS_LIB_PATH=$PWD/$f timeout 1500 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_gpu_fullsize.py} -x -q 2>&1 | tail -8; done > $OUT/${TAG}_partests.txt 2>&1; cat $OUT/${TAG}_partests.txt ;;
    ab)              # interleaved A/B timing of library builds (three runs each), per-kernel us / view
      for rep in 1 2 3; do for f in $(libs); do
        echo -n "$(basename $f) rep $rep: "
        GGS_LIB_PATH=$PWD/$f timeout 600 python bench.py --steps 30 --warmup 3 --cpu-views 0 --loop-views 0 --extra-configs 0 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print(d['value'], {a: round(b / d['roofline']['launch_views'] * 1e3, 2) for a, b in k.items()})"
      done; done > $OUT/${TAG}_ab.txt 2>&1; cat $OUT/${TAG}_ab.txt ;;
    prof)            # rocprofv3 passes of tools/profile_all.sh for each variant library: PASSES (default "sq lds"), PROF_ARGS (bench flags)
      for f in $(libs); do n=$(basename $f .so); GGS_LIB_PATH=$PWD/$f PASSES="${PASSES:-sq lds}" bash tools/profile_all.sh ${TAG}_$n $PROF_ARGS > $OUT/${TAG}_prof_$n.log 2>&1
        cat $OUT/prof_${TAG}_${n}_sq_counters.md $OUT/prof_${TAG}_${n}_lds_counters.md 2>/dev/null | grep -v "^$"; done ;;
    gputests)        timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/${TAG}_gputests.txt; cat $OUT/${TAG}_gputests.txt ;;
    seltests)        # a pytest selection (TESTS) against the product library
      timeout 2400 python -m pytest ${TESTS:-tests/test_gpu_graph_step.py} -x -q 2>&1 | tail -25 > $OUT/${TAG}_seltests.txt; cat $OUT/${TAG}_seltests.txt ;;
    graphstep)       # graph-replayed s2 iteration: it/s, three interleaved runs of the product library or of the variants in LIBS
      for i in 1 2 3; do for f in $( [ -z "$LIBS" ] && echo gaussian-garments_amd/csrc/libggsplat.so || libs ); do echo -n "$(basename $f): "; GGS_LIB_PATH=$PWD/$f timeout 600 python tools/profile_graph_step.py 256 2>&1 | tail -1; done; done > $OUT/${TAG}_graphstep.txt; cat $OUT/${TAG}_graphstep.txt ;;
    pipestep)        # sequential vs pipelined replay of the captured s2 iteration (tools/profile_graph_step.py [--pipelined]), three interleaved runs
      for i in 1 2 3; do timeout 300 python tools/profile_graph_step.py 256 2>&1 | tail -1; timeout 300 python tools/profile_graph_step.py 256 --pipelined 2>&1 | tail -1; done > $OUT/${TAG}_pipestep.txt; cat $OUT/${TAG}_pipestep.txt ;;
    sparsestep)      # captured s2 iteration with silhouette masks: sparse-mask loss pass against the plain one, sequential and pipelined
      for i in 1 2 3; do for a in "--silhouette --plain-loss" "--silhouette" "--silhouette --plain-loss --pipelined" "--silhouette --pipelined" "--pipelined"; do
        timeout 300 python tools/profile_graph_step.py 256 $a 2>&1 | tail -1; done; done > $OUT/${TAG}_sparsestep.txt; cat $OUT/${TAG}_sparsestep.txt ;;
    graphprof)       # kernel table of the graph-replayed s2 iteration (rocprofv3 --kernel-trace, last 60 periods)
      R=$PWD; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/gs -o g -- python $R/tools/profile_graph_step.py 64 > $R/$OUT/gs.log 2>&1)
      python tools/rocpd_summary.py $(find $OUT/gs -name "*.db" | head -1) --cycles 60 --anchor ${ANCHOR:-k_adam_multi} > $OUT/${TAG}_graph_step_kernels.md 2>&1
      grep "graphed s2 step" $OUT/gs.log >> $OUT/${TAG}_graph_step_kernels.md; rm -rf $OUT/gs; cat $OUT/${TAG}_graph_step_kernels.md ;;
    timeloss)        # fused photometric loss, us per 1080p view, for each variant library (LIBS) or the product (LIBS=product)
      for f in $( [ "$LIBS" = product ] && echo gaussian-garments_amd/csrc/libggsplat.so || libs ); do echo "--- $f"; for i in 1 2; do GGS_LIB_PATH=$PWD/$f timeout 300 python tools/dbg/time_loss.py 2>&1 | grep "^V="; done; done > $OUT/${TAG}_timeloss.txt; cat $OUT/${TAG}_timeloss.txt ;;
    autograd_floor)  timeout 300 python tools/dbg/autograd_floor.py > $OUT/${TAG}_autograd_floor.txt 2>&1; cat $OUT/${TAG}_autograd_floor.txt ;;
    pipesweep)       # serial vs two-stream pipelined step over launch-set sizes + the 20-view rank step (tools/dbg/pipeline_sweep.py)
      timeout 1500 python tools/dbg/pipeline_sweep.py ${REPS:-2} "$SEL" > $OUT/${TAG}_pipeline_sweep.txt 2>&1; cat $OUT/${TAG}_pipeline_sweep.txt ;;
    pipetrace)       # kernel trace of the serial and the pipelined step: which kernels were in flight together (tools/overlap_summary.py)
      R=$PWD; for mode in ${MODES:-0 1}; do
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/$OUT/pt$mode -o p -- python $R/bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 --steps 4 --warmup 2 --timing-only --chunk ${CHUNK:-40} --pipeline $mode $BENCH_ARGS > $R/$OUT/pt$mode.log 2>&1)
        { echo "## --pipeline $mode --chunk ${CHUNK:-40} $BENCH_ARGS (last ${LAST_MS:-40} ms of the trace)"; echo; python tools/overlap_summary.py $(find $OUT/pt$mode -name "*.db" | head -1) --last-ms ${LAST_MS:-40}; echo; grep -h '^{' $OUT/pt$mode.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench line under the tracer:', d['value'], 'views/s,', d['ms_per_step'], 'ms per step')"; echo; } >> $OUT/${TAG}_pipeline_overlap.md 2>&1
        rm -rf $OUT/pt$mode; done; cat $OUT/${TAG}_pipeline_overlap.md ;;
    timefwd)         # forward-only kernel times per view for each variant library (tools/dbg/time_fwd.py): prices the background stores
      for i in 1 2; do for f in $(libs); do GGS_LIB_PATH=$PWD/$f timeout 300 python tools/dbg/time_fwd.py ${VIEWS:-40} 2>&1 | tail -1; done; done > $OUT/${TAG}_timefwd.txt; cat $OUT/${TAG}_timefwd.txt ;;
    ec)              # two-wave (evaluator / compositor) latency kernels: parity first (short timeouts: a barrier bug hangs), then A/B timing
      { timeout 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q 2>&1 | tail -6
        for ec in 0 1 0 1; do echo -n "GGS_QUAD_EC=$ec "; GGS_QUAD_EC=$ec timeout 200 python tools/dbg/time_fwd.py 1 20 2>&1 | tail -1; done
        for ec in 0 1 0 1; do echo -n "GGS_QUAD_EC=$ec "; GGS_QUAD_EC=$ec timeout 300 python tools/profile_graph_step.py 256 2>&1 | tail -1; done; } > $OUT/${TAG}_ec.txt 2>&1; cat $OUT/${TAG}_ec.txt ;;
    ldspad)          # occupancy cap of the per-quadrant kernels through unused dynamic LDS (GGS_QUAD_LDS_PAD): forward time + graph step
      for pad in ${PADS:-0 8192 14336 17408 24576 36864}; do echo -n "GGS_QUAD_LDS_PAD=$pad "; GGS_QUAD_LDS_PAD=$pad timeout 200 python tools/dbg/time_fwd.py 1 20 2>&1 | tail -1
        echo -n "GGS_QUAD_LDS_PAD=$pad "; GGS_QUAD_LDS_PAD=$pad timeout 300 python tools/profile_graph_step.py 256 2>&1 | tail -1; done > $OUT/${TAG}_ldspad.txt 2>&1; cat $OUT/${TAG}_ldspad.txt ;;
    replicas)        # replica mode (R independent s2 registrations on R streams): aggregate rates, then the kernel overlap of R = 4 under the tracer
      { echo "## aggregate rates (unprofiled): python tools/dbg/replica_rate.py 128"; echo; timeout 600 python tools/dbg/replica_rate.py 128 2>&1 | grep -v amdgpu.ids; echo;
        R=$PWD; for rr in ${REPLICAS:-1 4}; do (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/$OUT/rep$rr -o p -- python $R/tools/dbg/replica_rate.py 48 --trace $rr > $R/$OUT/rep$rr.log 2>&1)
          echo "## R = $rr under rocprofv3 --kernel-trace (last ${LAST_MS:-30} ms of the trace)"; echo; python tools/overlap_summary.py $(find $OUT/rep$rr -name "*.db" | head -1) --last-ms ${LAST_MS:-30}; echo; grep "^R = " $OUT/rep$rr.log; echo; rm -rf $OUT/rep$rr; done; } > $OUT/${TAG}_replicas.md 2>&1; cat $OUT/${TAG}_replicas.md ;;
    stampstep)       # per-kernel microseconds of the captured s2 iteration from timestamps captured inside the graph (no profiler)
      for f in $( [ -z "$LIBS" ] && echo gaussian-garments_amd/csrc/libggsplat.so || libs ); do GGS_LIB_PATH=$PWD/$f timeout 300 python tools/dbg/stamp_graph_step.py ${N:-64} 2>&1 | grep -v amdgpu.ids | tail -3; done > $OUT/${TAG}_stampstep.txt; cat $OUT/${TAG}_stampstep.txt ;;
    knn)             timeout 300 python tools/dbg/time_knn.py > $OUT/${TAG}_knn.txt 2>&1; tail -2 $OUT/${TAG}_knn.txt ;;
    bench)           timeout 900 python bench.py $BENCH_ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; cat $OUT/${TAG}_bench.json ;;
    *) echo "unknown step $step" ;;
  esac
done
