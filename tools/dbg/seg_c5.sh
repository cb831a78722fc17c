cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 1 128" "1 1 128" "1 1 256" "1 0 128"; do
  set -- $cfg
  export GGS_SEG=$1 GGS_SEG_QUAD=$2 GGS_SEG_LEN=$3
  rm -rf /tmp/segp; rocprofv3 --kernel-trace --stats -d /tmp/segp -o t -- python $R/bench.py --cpu-views 0 --loop-views 0 --views 1 --chunk 1 --no-graph --steps 5 --warmup 2 --sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 > /tmp/segp.log 2>&1
  echo "== c5 single view GGS_SEG=$1 QUAD=$2 LEN=$3"
  python $R/tools/rocpd_summary.py $(find /tmp/segp -name '*.db' | head -1) | grep -E "seg_|render_" | cut -c1-100
done
