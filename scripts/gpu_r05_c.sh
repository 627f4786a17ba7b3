#!/bin/bash
# round 5, call C: parity subset + headline bench after a tree-kernel change
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
timeout 300 python __graft_entry__.py smoke > $OUT/c.smoke.log 2>&1; tail -n 1 $OUT/c.smoke.log
( time timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_tree.py} -x -q ${XD:--n 8} --durations=5 > $OUT/c.tests.log 2>&1 ) 2>&1 | grep real; tail -n 6 $OUT/c.tests.log
timeout 600 python bench.py $HEAD > $OUT/c.bench.json 2> $OUT/c.bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/c.bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','mean_trace_len','error_games')})
for rk in ('roofline','roofline_other'):
    print('  ', d[rk]['kernel'][:30], d[rk]['avg_launch_ms'], d[rk]['frac'])
print('  ', d['last_sim_phase_kcycles'])
PY
timeout 300 python scripts/wave_cycles.py $OUT/c.wave_cycles.npz 2> $OUT/c.wave.err | tail -n 1
