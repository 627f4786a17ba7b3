#!/bin/bash
# collector-only launches beside the evaluator (an experiment: needs docs/experiments/r05_collectors_beside_the_evaluator.patch applied -
# tm_store::gc_side_cycles, bench.py --gc-side-cycles): steady-state windows under both nets, side:slice pairs
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
run() {  # name, args
  local name=$1; shift
  timeout 300 python bench.py --others none --no-cpu-baseline --steady-steps 0 --warmup 75 --steps 20 "$@" > $OUT/s.$name.json 2> $OUT/s.$name.err
  python - <<PY
import json
d=json.load(open('$OUT/s.$name.json'))
g=d['gc']; rk=[d[r] for r in ('roofline','roofline_other') if d[r]['kernel'].startswith('k_sim')][0]; nn=[d[r] for r in ('roofline','roofline_other') if not d[r]['kernel'].startswith('k_sim')][0]
print('%-26s ms/move %6.1f  exp/s %.2fM  waited/coll %5.1f  catchup/move %5.1f  gc-only %d  tree %.1f us  nn %.1f us  err %d' % ('$name', d['ms_per_step'], d['value']/1e6, g['launches_per_collection'] or 0, g['catchup_launches_per_move'], g['collector_only_launches'], 1e3*rk['avg_launch_ms'], 1e3*nn['avg_launch_ms'], d['error_games']))
PY
}
CK=tetris_mcts_amd/checkpoints/value_net_online_r05.pt
for spec in "$@"; do
  IFS=: read side slice <<< "$spec"
  run trained_side${side}_slice${slice} --checkpoint $CK --gc-side-cycles $side --gc-slice-cycles $slice
  run random_side${side}_slice${slice} --gc-side-cycles $side --gc-slice-cycles $slice
done
