#!/bin/bash
# round 5, call D: collector changes (age order, grid guard, catch-up over the owing games) + the default bench line
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
timeout 300 python __graft_entry__.py smoke > $OUT/d.smoke.log 2>&1; tail -n 1 $OUT/d.smoke.log
( time timeout 1500 python -m pytest $TESTS -x -q -s --durations=8 > $OUT/d.tests.log 2>&1 ) 2>&1 | grep real; grep "launches waited" $OUT/d.tests.log; tail -n 14 $OUT/d.tests.log
[ -n "$NOBENCH" ] && exit 0
( time timeout 900 python bench.py > $OUT/d.bench.json 2> $OUT/d.bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/d.bench.json'))
print({k:v for k,v in d.items() if isinstance(v,(int,float)) and not isinstance(v,bool)})
print('steady gc', d['steady_state']['gc'])
t=d.get('trained_net',{}); print('trained', {k:t.get(k) for k in ('value','ms_per_step','mean_trace_len','lines_per_1000_moves','mean_lines_all_episodes_under_way','error')}); print('trained steady', {k:v for k,v in t.get('steady_state',{}).items() if not isinstance(v,(dict,str))})
PY
