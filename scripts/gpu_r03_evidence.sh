#!/bin/bash
# Regenerates the round-3 evidence under gpurun_out/ev.* (copied to profiles/r03_* afterwards):
#   scripts/gpu_r03_evidence.sh [bench|prof|lp|suite ...]     default: all parts
#   bench  the driver's command line (python bench.py): headline + steady-state window + CPU baseline
#   prof   rocprofv3 kernel trace and PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters) of the headline window; kernel trace of
#          the steady-state window
#   lp     BASELINE configs[2] (ValueSimLP) at the default 5 + 20 protocol, its kernel trace and PMC passes
#   suite  the whole -m gpu suite + smoke
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
PARTS=${@:-bench prof lp suite}
HEAD="--no-cpu-baseline --steady-steps 0"
prof_kt() {   # name, extra bench args..., prints the per-kernel table of the last N launches
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/ev.$name.json 2> $R/$OUT/ev.$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/ev.kernel_stats_$name.csv --last $last; head -n 6 $OUT/ev.kernel_stats_$name.csv | cut -c1-160
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
  grep "k_sim_step\|k_vn_\|k_fc_out" $OUT/ev.pmc_traffic_$name.csv
}
for p in $PARTS; do case $p in
bench)
  ( time timeout 900 python bench.py > $OUT/ev.bench.json 2> $OUT/ev.bench.err ) 2>&1 | grep real
  python -c "
import json; d=json.load(open('$OUT/ev.bench.json')); print({k:d.get(k) for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline_other']['frac'], d['roofline']['traffic'], d['roofline_other']['traffic']); print(d['steady_state']['ms_per_step'], d['steady_state']['gc']); print(d['cpu_baseline']['cores'], d['cpu_baseline']['value'])" ;;
prof)
  prof_kt head 10000 $HEAD
  prof_pmc head 10000 $HEAD
  cd /tmp; rm -rf /tmp/p_sq
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/p_sq -- python $R/bench.py $HEAD --steps 4 > /dev/null 2> $R/$OUT/ev.sq.err; echo "sq rc=$?"
  cd $R; python scripts/pmc_traffic.py $OUT/ev.pmc_sq.json $OUT/ev.pmc_sq.csv /tmp/p_sq --last 2000 --workload-key "head (4 timed moves)" --command "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace -- python bench.py $HEAD --steps 4 (values are raw counter sums per dispatch, not KiB)"
  grep "k_sim_step" $OUT/ev.pmc_sq.csv
  prof_kt steady 10000 --no-cpu-baseline --steady-steps 0 --warmup 75 --steps 20 ;;
lp)
  prof_kt lp 10000 --agent ValueSimLP $HEAD
  prof_pmc lp 10000 --agent ValueSimLP $HEAD
  timeout 900 python bench.py --agent ValueSimLP --no-cpu-baseline --steady-steps 0 > $OUT/ev.bench_lp.json 2> $OUT/ev.bench_lp.err; echo "lp bench rc=$?"
  python -c "
import json; d=json.load(open('$OUT/ev.bench_lp.json')); print({k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec','mean_trace_len')}); print(d['roofline']['kernel'][:30], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'])" ;;
suite)
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/ev.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/ev.smoke.log | cut -c1-300
  timeout 1500 python -m pytest tests -m gpu -q -n 6 --maxfail=12 --durations=6 > $OUT/ev.pytest.log 2>&1
  echo "pytest rc=$?"; tail -n 14 $OUT/ev.pytest.log | cut -c1-300 ;;
esac; done
