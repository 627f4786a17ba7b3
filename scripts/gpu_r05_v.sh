#!/bin/bash
# bench the headline with variant libraries: scripts/gpu_r05_v.sh name=path ...
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
HEAD="--no-cpu-baseline --steady-steps 0 --others none --warmup 5 --steps ${STEPS:-10}"
for v in "$@"; do
  name=${v%%=*}; lib=${v#*=}
  TETRIS_MCTS_LIB=$R/$lib timeout 300 python bench.py $HEAD > $OUT/v.$name.json 2> $OUT/v.$name.err; echo "$name rc=$?"
  python - <<PY
import json
d=json.load(open('$OUT/v.$name.json'))
print('$name', {k:round(d.get(k),3) for k in ('value','ms_per_step','mean_trace_len')}, [ (d[rk]['kernel'][:10], round(d[rk]['avg_launch_ms'],4)) for rk in ('roofline','roofline_other')], {k:round(v,1) for k,v in d['last_sim_phase_kcycles'].items() if k.startswith('CYC')})
PY
done
