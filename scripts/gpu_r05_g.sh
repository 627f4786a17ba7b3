#!/bin/bash
# the final collector configuration (128 collectors, five marking workgroups per game): speculative threshold / marking allowance once more, random-init steady state
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
for spec in "$@"; do
  IFS=: read nodes slice <<< "$spec"
  timeout 300 python bench.py --others none --no-cpu-baseline --steady-steps 0 --warmup 75 --steps 20 --gc-spec-nodes $nodes --gc-slice-cycles $slice > $OUT/h.s${nodes}_t${slice}.json 2> /dev/null
  python - <<PY
import json
d=json.load(open('$OUT/h.s${nodes}_t${slice}.json')); g=d['gc']
print('spec %5d slice %6d  ms/move %6.1f  waited/coll %5.1f  catchup/move %5.1f' % ($nodes, $slice, d['ms_per_step'], g['launches_per_collection'] or 0, g['catchup_launches_per_move']))
PY
done
