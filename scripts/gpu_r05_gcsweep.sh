#!/bin/bash
# the collector's parameters on the steady-state window (moves 76-95) UNDER THE TRAINED NET: every simulation expands, the pools
# fill twice as fast as under the random-init net and twice as many collections are under way
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
CK=tetris_mcts_amd/checkpoints/value_net_online_r05.pt
run() {  # name, args
  local name=$1; shift
  timeout 300 python bench.py --checkpoint $CK --others none --no-cpu-baseline --warmup 75 --steps 20 --steady-steps 0 "$@" > $OUT/g.$name.json 2> $OUT/g.$name.err
  python - <<PY
import json
d=json.load(open('$OUT/g.$name.json'))
g=d['gc']; rk=[d[r] for r in ('roofline','roofline_other') if d[r]['kernel'].startswith('k_sim')][0]
print('%-22s ms/move %6.1f  exp/s %.2fM  waited/coll %5.1f  catchup/move %5.1f (eq %.1f)  gc-only %d  restarts %d  tree %.1f us  lines/1000 %.1f' % ('$name', d['ms_per_step'], d['value']/1e6, g['launches_per_collection'] or 0, g['catchup_launches_per_move'], g['catchup_full_launch_equivalents_per_move'], g['collector_only_launches'], g['trees_restarted_pool_outgrown'], 1e3*rk['avg_launch_ms'], d['lines_per_1000_moves']))
PY
}
for spec in "$@"; do
  IFS=: read cost nodes slice wgs <<< "$spec"
  run c${cost}_s${nodes}_t${slice}_w${wgs:-64} --gc-cost-units $cost --gc-spec-nodes $nodes --gc-slice-cycles $slice --gc-collectors ${wgs:-64}
done
