#!/bin/bash
# round 4, GPU call B: fc1 with four workgroups per tile, the distributional walk's prediction and backup prefetch; the driver's
# bench command; kernel traces and PMC passes of the headline window
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/b.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/b.smoke.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_valuenet.py tests/test_gpu_dist_agent.py tests/test_gpu_tree.py tests/test_gpu_benched_regime.py -m gpu -q -x -n 4 \
   -k "valuenet or dist_agent or hip_head or leaf_parallel or value_net_in_the_loop or reference_golden_runs or 500_sims" > $OUT/b.new.log 2>&1; echo "new rc=$?"; tail -n 12 $OUT/b.new.log | cut -c1-600
timeout 900 python -m pytest tests -m gpu -q -x -k "sampled_games or benchmarked_dist" > $OUT/b.big.log 2>&1; echo "big rc=$?"; tail -n 6 $OUT/b.big.log | cut -c1-600
( time timeout 900 python bench.py > $OUT/b.bench.json 2> $OUT/b.bench.err ) 2>&1 | grep real
python - <<PY
import json
d=json.load(open('$OUT/b.bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec')}, d['requests']['fraction_not_posted'])
for rk in ('roofline','roofline_other'): print(d[rk]['kernel'][:40], d[rk]['avg_launch_ms'], d[rk]['frac'])
ss=d['steady_state']; print('steady', ss['value'], ss['ms_per_step'], ss['gc'], ss['tree_kernel_ms'], ss['value_net_ms'], ss['requests']['fraction_not_posted'])
cb=d['cpu_baseline']; print('cpu', cb['kind'], cb['cores'], cb['value'], cb.get('one_core'))
for k,v in d.get('other_configs',{}).items():
    print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','error')}, [(v[rk]['avg_launch_ms'], v[rk]['frac']) for rk in ('roofline','roofline_other') if rk in v])
    if 'cpu_baseline' in v: print('   cpu', {kk:v['cpu_baseline'].get(kk) for kk in ('kind','cores','value','one_core','oracle_port_one_core')})
PY
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
prof_kt() {   # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/b.kt_$name.json 2> $R/$OUT/b.kt_$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/b.kernel_stats_$name.csv --last $last; head -n 5 $OUT/b.kernel_stats_$name.csv | cut -c1-60,150-400
}
prof_pmc() {  # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_${name}_f /tmp/p_${name}_w
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_${name}_f -- python $R/bench.py "$@" > /dev/null 2> $R/$OUT/b.${name}_fetch.err; echo "$name fetch rc=$?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_${name}_w -- python $R/bench.py "$@" > /dev/null 2> $R/$OUT/b.${name}_write.err; echo "$name write rc=$?"
  cd $R
  KEY=$(python -c "import json;print(json.load(open('$OUT/b.kt_$name.json'))['config']['workload_key'])")
  python scripts/pmc_traffic.py $OUT/b.pmc_traffic_$name.json $OUT/b.pmc_traffic_$name.csv /tmp/p_${name}_f /tmp/p_${name}_w --last $last --workload-key "$KEY" \
    --command "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py $* (one pass per counter; averaged over the last $last launches of every kernel = the timed window)"
  grep "k_sim_step\|k_vn_\|k_dn_" $OUT/b.pmc_traffic_$name.csv
}
prof_kt head 10000 $HEAD
prof_pmc head 10000 $HEAD
prof_kt steady 10000 --no-cpu-baseline --steady-steps 0 --others none --warmup 75 --steps 20
prof_kt lp 10000 --agent ValueSimLP $HEAD
prof_kt dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD
