cd $GRAFT_REPO_ROOT
echo "== default"; bash tools/dbg/ab_libs.sh
echo "== one view per launch"; BENCH_ARGS="--views 1 --chunk 1" bash tools/dbg/ab_libs.sh
echo "== graph step"
for rep in 1 2; do for f in gaussian-garments_amd/csrc/variants/*.so; do echo -n "$(basename $f): "; GGS_LIB_PATH=$PWD/$f python tools/profile_graph_step.py 128 | tail -1; done; done
