cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in gaussian-garments_amd/csrc/variants/*.so; do echo "== $(basename $f)"; GGS_LIB_PATH=$PWD/$f python tools/bench_next_rows.py 2>/dev/null | grep "f1" ; GGS_LIB_PATH=$PWD/$f python tools/profile_graph_step.py 128 2>&1 | tail -1; done; done
