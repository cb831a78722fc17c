#!/bin/bash
# Everything profiles/ holds for one round, from ONE build, in one GPU-box call:  tools/profile_round.sh r03
# (run it through gpurun; the summaries land in gpurun_out/prof_<round>_*, the default bench line in
# gpurun_out/<round>_bench_default.json -- copy both sets into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-r03}
cd $R
WORKLOAD="100000 1920 1080 0" bash tools/profile_all.sh ${N}_sh0 --chunk 32
WORKLOAD="100000 1920 1080 3" PASSES="trace sq valu fetch write" bash tools/profile_all.sh ${N}_sh3 --sh-degree 3 --chunk 32
WORKLOAD="500000 3840 2160 3" PASSES="trace sq valu fetch write" bash tools/profile_all.sh ${N}_c5 --sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 --chunk 16 --views 32
ONLY_TRACE=1 bash tools/profile_all.sh ${N}_v20 --views 20 --chunk 20
# the launch shape of the DEFAULT bench line (40 views per launch, launch sets pipelined over two streams), and the same with
# serial launch sets: the per-launch durations of roofline.launch_ms
ONLY_TRACE=1 bash tools/profile_all.sh ${N}_default
ONLY_TRACE=1 bash tools/profile_all.sh ${N}_serial --pipeline 0
WORKLOAD="100000 1920 1080 0" PMC_VIEWS=1 PASSES="trace sq fetch write" bash tools/profile_all.sh ${N}_v1 --views 1 --chunk 1 --no-graph
# graph-replayed s2 step: where one iteration's GPU time goes
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gs -o g -- python $R/tools/profile_graph_step.py 64 > $R/gpurun_out/gs.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/gs -name "*.db" | head -1) --cycles 60 --anchor k_adam_multi > $R/gpurun_out/prof_${N}_graph_step_kernels.md 2>&1
grep "graphed s2 step" $R/gpurun_out/gs.log >> $R/gpurun_out/prof_${N}_graph_step_kernels.md
rm -rf $R/gpurun_out/gs
( cd $R; echo; echo "Unprofiled (two runs each of \`python tools/profile_graph_step.py 256\`, \`... 256 --pipelined\` and \`... 256 --pipelined --silhouette [--plain-loss]\`):"; for i in 1 2; do python tools/profile_graph_step.py 256 | tail -1; python tools/profile_graph_step.py 256 --pipelined | tail -1; python tools/profile_graph_step.py 256 --pipelined --silhouette | tail -1; python tools/profile_graph_step.py 256 --pipelined --silhouette --plain-loss | tail -1; done ) >> $R/gpurun_out/prof_${N}_graph_step_kernels.md 2>/dev/null
# serial / whole-forward pipelined / stage-granular pipelined step: which kernels were in flight together (tools/overlap_summary.py)
( cd $R; TAG=prof_${N} MODES="0 1 2" CHUNK=40 bash tools/dbg/job.sh pipetrace > /dev/null 2>&1 )
# the 20-view rank step of an 8-GPU run, unprofiled, on one GPU + serial vs pipelined launch sets
( cd $R; TAG=prof_${N} REPS=2 SEL="chunk40|chunk80 serial  m2d1|rank:" bash tools/dbg/job.sh pipesweep > /dev/null 2>&1 )
# the default bench line needs the traffic / VALU collections of THIS build in profiles/: copy them in on the box first
cd $R
for f in gpurun_out/prof_${N}_*; do cp $f profiles/$(basename $f | sed 's/^prof_//'); done
python bench.py > gpurun_out/${N}_bench_default.json 2> gpurun_out/${N}_bench_default.err
python tools/bench_next_rows.py > gpurun_out/prof_${N}_next_rows.md 2> gpurun_out/${N}_next_rows.err || true
python tools/bench_stylegan_ops.py > gpurun_out/prof_${N}_stylegan_ops.md 2> gpurun_out/${N}_stylegan_ops.err || true
python tools/isa_audit.py > gpurun_out/prof_${N}_isa_audit.md 2> gpurun_out/${N}_isa_audit.err || true
tail -c 400 gpurun_out/${N}_bench_default.json
