#!/bin/bash
# steady-state windows under both nets with variant libraries: scripts/gpu_r05_f.sh name=path ...
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
run() {  # name, lib, args
  local name=$1 lib=$2; shift 2
  TETRIS_MCTS_LIB=$R/$lib timeout 300 python bench.py --others none --no-cpu-baseline --steady-steps 0 "$@" > $OUT/f.$name.json 2> $OUT/f.$name.err
  python - <<PY
import json
d=json.load(open('$OUT/f.$name.json'))
g=d['gc']; rk=[d[r] for r in ('roofline','roofline_other') if d[r]['kernel'].startswith('k_sim')][0]
print('%-22s ms/move %6.1f  exp/s %.2fM  waited/coll %5.1f  catchup/move %5.1f  gc-only %d  tree %.1f us' % ('$name', d['ms_per_step'], d['value']/1e6, g['launches_per_collection'] or 0, g['catchup_launches_per_move'], g['collector_only_launches'], 1e3*rk['avg_launch_ms']))
PY
}
for v in "$@"; do
  name=${v%%=*}; lib=${v#*=}
  run ${name}_random $lib --warmup 75 --steps 20
  run ${name}_trained $lib --checkpoint tetris_mcts_amd/checkpoints/value_net_online_r05.pt --warmup 75 --steps 20
done
