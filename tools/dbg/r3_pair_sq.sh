cd $GRAFT_REPO_ROOT
for pm in 0 1; do
  export GGS_PAIR=$pm
  WORKLOAD="100000 1920 1080 0" PMC_VIEWS=1 PASSES="sq" bash tools/profile_all.sh pair$pm --views 1 --chunk 1 --no-graph > /dev/null 2>&1
  echo "== GGS_PAIR=$pm"; grep -E "render_fwd|render_bwd|^\| kernel" gpurun_out/prof_pair${pm}_sq_counters.md
done
