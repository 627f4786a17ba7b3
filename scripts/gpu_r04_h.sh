#!/bin/bash
# round 4, GPU call H: fc1 in two tilings (32-state tiles with a deep weight ring, 64-state tiles for the large launches)
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/h.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/h.smoke.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_valuenet.py tests/test_gpu_dist_agent.py tests/test_gpu_tree.py -m gpu -q -x -n 4 \
   -k "valuenet or hip_head or with_the_hip_head or value_net_in_the_loop" > $OUT/h.new.log 2>&1; echo "new rc=$?"; tail -n 6 $OUT/h.new.log | cut -c1-600
show() { python - <<PY
import json
d=json.load(open('$1'))
print('$1', {k:d.get(k) for k in ('value','ms_per_step')}, d['requests']['fraction_not_posted'])
for rk in ('roofline','roofline_other'):
    if rk in d: print('  ', d[rk]['kernel'][:40], d[rk]['avg_launch_ms'], d[rk]['frac'])
if 'steady_state' in d:
    ss=d['steady_state']; print('   steady', ss['value'], ss['ms_per_step'], ss['gc']['catchup_launches_per_move'], ss['tree_kernel_ms'], ss['value_net_ms'], ss.get('episodes_finished'), ss.get('lines_cleared_per_episode'))
for k,v in d.get('other_configs',{}).items():
    print('  ', k, {kk:v.get(kk) for kk in ('value','ms_per_step','error')}, [(v[rk]['avg_launch_ms'], v[rk]['frac']) for rk in ('roofline','roofline_other') if rk in v])
PY
}
timeout 600 python bench.py --no-cpu-baseline > $OUT/h.bench.json 2> $OUT/h.bench.err; echo "bench rc=$?"; show $OUT/h.bench.json
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
prof_kt() {   # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/h.kt_$name.json 2> $R/$OUT/h.kt_$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/h.kernel_stats_$name.csv --last $last; head -n 5 $OUT/h.kernel_stats_$name.csv | cut -c1-60,150-400
}
prof_kt head 10000 $HEAD
prof_kt lp 10000 --agent ValueSimLP $HEAD
prof_kt dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD
