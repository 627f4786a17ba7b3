#!/bin/bash
# round 4, GPU call C: fc1 (parallel slot resolve, activation prefetch two chunks ahead), the distributional agent's online leg,
# sub-batches on two streams once more
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/c.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/c.smoke.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_valuenet.py tests/test_gpu_dist_agent.py tests/test_gpu_tree.py tests/test_gpu_benched_regime.py -m gpu -q -x -n 4 \
   -k "valuenet or dist_agent or hip_head or dist_online or value_net_in_the_loop or 500_sims or replay_harvest" > $OUT/c.new.log 2>&1; echo "new rc=$?"; tail -n 12 $OUT/c.new.log | cut -c1-600
timeout 600 python -m pytest tests -m gpu -q -x -k "sampled_games or benchmarked_dist" > $OUT/c.big.log 2>&1; echo "big rc=$?"; tail -n 4 $OUT/c.big.log | cut -c1-600
show() { python - <<PY
import json
d=json.load(open('$1'))
print('$1', {k:d.get(k) for k in ('value','ms_per_step')}, d['requests']['fraction_not_posted'])
for rk in ('roofline','roofline_other'):
    if rk in d: print('  ', d[rk]['kernel'][:40], d[rk]['avg_launch_ms'], d[rk]['frac'])
if 'steady_state' in d:
    ss=d['steady_state']; print('   steady', ss['value'], ss['ms_per_step'], ss['gc']['catchup_launches_per_move'], ss['tree_kernel_ms'], ss['value_net_ms'])
for k,v in d.get('other_configs',{}).items():
    print('  ', k, {kk:v.get(kk) for kk in ('value','ms_per_step','error')}, [(v[rk]['avg_launch_ms'], v[rk]['frac']) for rk in ('roofline','roofline_other') if rk in v])
PY
}
timeout 600 python bench.py --no-cpu-baseline > $OUT/c.bench.json 2> $OUT/c.bench.err; echo "bench rc=$?"; show $OUT/c.bench.json
for NS in 2 4; do
timeout 300 python bench.py --no-cpu-baseline --others none --steady-steps 0 --split $NS > $OUT/c.bench_split$NS.json 2> $OUT/c.bench_split$NS.err; echo "split $NS rc=$?"; show $OUT/c.bench_split$NS.json
done
timeout 300 python bench.py --no-cpu-baseline --others none --steady-steps 0 --split 2 --agent ValueSimLP > $OUT/c.bench_lp_split2.json 2> $OUT/c.bench_lp_split2.err; echo "lp split 2 rc=$?"; show $OUT/c.bench_lp_split2.json
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
prof_kt() {   # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/c.kt_$name.json 2> $R/$OUT/c.kt_$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/c.kernel_stats_$name.csv --last $last; head -n 5 $OUT/c.kernel_stats_$name.csv | cut -c1-60,150-400
}
prof_kt head 10000 $HEAD
prof_kt lp 10000 --agent ValueSimLP $HEAD
prof_kt dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD
timeout 240 python scripts/selfplay_online.py --agent DistValueSim --minutes 2.5 --games 512 --sims 200 --max-nodes 30000 --train-every 50 --min-visits 20 --out $OUT/c.online_dist.jsonl > $OUT/c.online_dist.log 2>&1; echo "online dist rc=$?"; tail -n 3 $OUT/c.online_dist.log | cut -c1-700
