#!/bin/bash
# Regenerates the round-5 evidence under gpurun_out/ev.* (copied to profiles/r05_* afterwards):
#   scripts/gpu_r04_evidence.sh [bench|prof|lp|dist|suite|online ...]     default: bench prof lp dist suite
#   bench  the driver's command line (python bench.py): headline + steady-state window + the other configs + CPU baselines
#   prof   rocprofv3 kernel trace and PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters) of the headline window; kernel trace of
#          the steady-state window
#   lp     BASELINE configs[2] (ValueSimLP): bench line, kernel trace, PMC passes
#   dist   BASELINE configs[4] (DistValueSim, 1000 sims/move): bench line, kernel trace, PMC passes
#   suite  the whole -m gpu suite (the tests that hold a 4096-game store one after the other) + smoke
#   trained the steady-state window under the committed checkpoint (the regime self-play lives in): bench line + kernel trace
#   online the learning curve + checkpoint: scripts/gpu_r05_train.sh
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
PARTS=${@:-bench prof lp dist suite}
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
prof_kt() {   # name, last, bench args...: per-kernel table of the last N launches
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/ev.$name.json 2> $R/$OUT/ev.$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/ev.kernel_stats_$name.csv --last $last; head -n 5 $OUT/ev.kernel_stats_$name.csv | cut -c1-60,150-400
}
prof_pmc() {  # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_${name}_f /tmp/p_${name}_w
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_${name}_f -- python $R/bench.py "$@" > /dev/null 2> $R/$OUT/ev.${name}_fetch.err; echo "$name fetch rc=$?"
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_${name}_w -- python $R/bench.py "$@" > /dev/null 2> $R/$OUT/ev.${name}_write.err; echo "$name write rc=$?"
  cd $R
  KEY=$(python -c "import json;print(json.load(open('$OUT/ev.$name.json'))['config']['workload_key'])")
  python scripts/pmc_traffic.py $OUT/ev.pmc_traffic_$name.json $OUT/ev.pmc_traffic_$name.csv /tmp/p_${name}_f /tmp/p_${name}_w --last $last --workload-key "$KEY" \
    --command "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py $* (one pass per counter; averaged over the last $last launches of every kernel = the timed window)"
  grep "k_sim_step\|k_vn_\|k_dn_" $OUT/ev.pmc_traffic_$name.csv
}
show() { python - <<PY
import json
d=json.load(open('$1'))
print('$1', {k:d.get(k) for k in ('value','ms_per_step')}, d['requests']['fraction_not_posted'])
for rk in ('roofline','roofline_other'):
    if rk in d: print('  ', d[rk]['kernel'][:40], d[rk]['avg_launch_ms'], d[rk]['frac'], d[rk]['traffic'])
if 'steady_state' in d:
    ss=d['steady_state']; print('   steady', ss['value'], ss['ms_per_step'], ss['gc'], ss['tree_kernel_ms'], ss['value_net_ms'], ss.get('episodes_finished'), ss.get('lines_cleared_per_episode'))
if 'cpu_baseline' in d:
    cb=d['cpu_baseline']; print('   cpu', cb.get('kind'), cb.get('cores'), cb.get('value'), cb.get('one_core'))
for k,v in d.get('other_configs',{}).items():
    print('  ', k, {kk:v.get(kk) for kk in ('value','ms_per_step','error')}, [(v[rk]['avg_launch_ms'], v[rk]['frac']) for rk in ('roofline','roofline_other') if rk in v])
    if 'cpu_baseline' in v: print('      cpu', {kk:v['cpu_baseline'].get(kk) for kk in ('kind','cores','value','one_core','oracle_port_one_core')})
PY
}
cd $R
for p in $PARTS; do case $p in
bench)
  ( time timeout 900 python bench.py > $OUT/ev.bench.json 2> $OUT/ev.bench.err ) 2>&1 | grep real
  show $OUT/ev.bench.json ;;
prof)
  prof_kt head 10000 $HEAD
  prof_pmc head 10000 $HEAD
  cd /tmp; rm -rf /tmp/p_sq
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sq -- python $R/bench.py $HEAD --warmup 1 --steps 2 > /dev/null 2> $R/$OUT/ev.sq.err; echo "sq rc=$?"
  cd $R; python scripts/pmc_traffic.py $OUT/ev.pmc_sq.json $OUT/ev.pmc_sq.csv /tmp/p_sq --last 1000 --workload-key "head (2 timed moves)" --command "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- python bench.py $HEAD --warmup 1 --steps 2 (raw counter sums per dispatch, not KiB)"
  grep "k_sim_step\|k_vn_" $OUT/ev.pmc_sq.csv | cut -c1-200
  prof_kt steady 10000 --no-cpu-baseline --steady-steps 0 --others none --warmup 75 --steps 20 ;;
lp)
  prof_kt lp 10000 --agent ValueSimLP $HEAD
  prof_pmc lp 10000 --agent ValueSimLP $HEAD
  timeout 900 python bench.py --agent ValueSimLP --no-cpu-baseline --others none --steady-steps 0 > $OUT/ev.bench_lp.json 2> $OUT/ev.bench_lp.err; echo "lp bench rc=$?"; show $OUT/ev.bench_lp.json ;;
dist)
  prof_kt dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD
  prof_pmc dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD
  timeout 900 python bench.py --agent DistValueSim --sims 1000 --warmup 2 --steps 5 --no-cpu-baseline --others none --steady-steps 0 > $OUT/ev.bench_dist.json 2> $OUT/ev.bench_dist.err; echo "dist bench rc=$?"; show $OUT/ev.bench_dist.json ;;
trained)
  CK=tetris_mcts_amd/checkpoints/value_net_online_r05.pt
  prof_kt trained_steady 10000 --checkpoint $CK --no-cpu-baseline --others none --steady-steps 0 --warmup 75 --steps 20
  timeout 600 python bench.py --checkpoint $CK --no-cpu-baseline --others none > $OUT/ev.bench_trained.json 2> $OUT/ev.bench_trained.err; echo "trained bench rc=$?"; show $OUT/ev.bench_trained.json ;;
suite)
  # one serial pytest process, as the driver runs it
  ( time timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $OUT/ev.pytest.log 2>&1 ) 2>&1 | grep real
  echo "pytest rc=$?"; tail -n 24 $OUT/ev.pytest.log | cut -c1-200
  timeout 300 python __graft_entry__.py smoke > $OUT/ev.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/ev.smoke.log | cut -c1-300 ;;
esac; done
