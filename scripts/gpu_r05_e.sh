#!/bin/bash
# steady-state windows under both nets after a collector change
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
run() {  # name, args
  local name=$1; shift
  timeout 300 python bench.py --others none --no-cpu-baseline --steady-steps 0 "$@" > $OUT/e.$name.json 2> $OUT/e.$name.err
  python - <<PY
import json
d=json.load(open('$OUT/e.$name.json'))
g=d['gc']; rk=[d[r] for r in ('roofline','roofline_other') if d[r]['kernel'].startswith('k_sim')][0]
print('%-22s ms/move %6.1f  exp/s %.2fM  waited/coll %5.1f  catchup/move %5.1f (eq %.1f)  gc-only %d  restarts %d  tree %.1f us' % ('$name', d['ms_per_step'], d['value']/1e6, g['launches_per_collection'] or 0, g['catchup_launches_per_move'], g['catchup_full_launch_equivalents_per_move'], g['collector_only_launches'], g['trees_restarted_pool_outgrown'], 1e3*rk['avg_launch_ms']))
PY
}
run steady_random --warmup 75 --steps 20
run steady_trained --checkpoint tetris_mcts_amd/checkpoints/value_net_online_r05.pt --warmup 75 --steps 20
run steady_lp --agent ValueSimLP --warmup 75 --steps 20
