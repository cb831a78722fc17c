R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gs -o g -- python $R/tools/profile_graph_step.py 64 > $R/gpurun_out/gs.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/gs -name "*.db" | head -1) --cycles 60 --anchor k_adam_multi > $R/gpurun_out/graph_step_kernels.md 2>&1
grep "graphed s2 step" $R/gpurun_out/gs.log >> $R/gpurun_out/graph_step_kernels.md
rm -rf $R/gpurun_out/gs
cd $R; python tools/profile_graph_step.py 256 | tail -1
python tools/profile_graph_step.py 256 | tail -1
