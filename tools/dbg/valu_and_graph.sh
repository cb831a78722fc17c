#!/bin/bash
# VALU busy / lane-activity counters for the config-2 workload + a kernel trace of the graph-replayed s2 step
R=$GRAFT_REPO_ROOT
WORKLOAD="100000 1920 1080 0" PASSES=valu bash $R/tools/profile_all.sh r02_sh0 --chunk 32
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gs -o g -- python $R/tools/profile_graph_step.py 64 > $R/gpurun_out/gs.log 2>&1
DB=$(find $R/gpurun_out/gs -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/r02_graph_step_kernels.md 2>&1
rm -rf $R/gpurun_out/gs
tail -3 $R/gpurun_out/gs.log
cat $R/gpurun_out/prof_r02_sh0_valu.md
head -50 $R/gpurun_out/r02_graph_step_kernels.md | cut -c1-150
