#!/bin/bash
# round 4, GPU call A: the new paths first (dist head, dense request list, needed-only requests), then the suite and bench lines
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/a.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/a.smoke.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_valuenet.py tests/test_gpu_dist_agent.py -m gpu -q -x -k "not benchmarked_dist" > $OUT/a.new.log 2>&1; echo "new rc=$?"; tail -n 25 $OUT/a.new.log | cut -c1-600
timeout 1500 python -m pytest tests -m gpu -q -n 6 --maxfail=20 --durations=8 -k "not (4096 or benchmarked_dist)" > $OUT/a.pytest.log 2>&1; echo "pytest rc=$?"; tail -n 40 $OUT/a.pytest.log | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 -k "4096 or benchmarked_dist" -s > $OUT/a.big.log 2>&1; echo "big rc=$?"; tail -n 40 $OUT/a.big.log | cut -c1-600
for A in ValueSim ValueSimLP; do
  timeout 600 python bench.py --agent $A --no-cpu-baseline --steady-steps 0 > $OUT/a.bench_$A.json 2> $OUT/a.bench_$A.err; echo "bench $A rc=$?"
  python -c "
import json; d=json.load(open('$OUT/a.bench_$A.json')); print({k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec','sims_per_sec')}); print(d['roofline']['kernel'][:40], d['roofline']['avg_launch_ms'], d['roofline']['frac']); print(d['roofline_other']['kernel'][:40], d['roofline_other']['avg_launch_ms'], d['roofline_other']['frac'])"
done
timeout 600 python bench.py --agent DistValueSim --sims 1000 --warmup 2 --steps 5 --no-cpu-baseline --steady-steps 0 > $OUT/a.bench_dist.json 2> $OUT/a.bench_dist.err; echo "bench dist rc=$?"
python -c "
import json; d=json.load(open('$OUT/a.bench_dist.json')); print({k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec','sims_per_sec')}); print(d['roofline']['kernel'][:40], d['roofline']['avg_launch_ms'], d['roofline']['frac']); print(d['roofline_other']['kernel'][:40], d['roofline_other']['avg_launch_ms'], d['roofline_other']['frac'])"
prof_kt() {   # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/a.kt_$name.json 2> $R/$OUT/a.kt_$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/a.kernel_stats_$name.csv --last $last; head -n 8 $OUT/a.kernel_stats_$name.csv | cut -c1-200
}
prof_kt vs 2500 --no-cpu-baseline --steady-steps 0 --warmup 5 --steps 5
prof_kt lp 2500 --agent ValueSimLP --no-cpu-baseline --steady-steps 0 --warmup 5 --steps 5
prof_kt dist 3000 --agent DistValueSim --sims 1000 --no-cpu-baseline --steady-steps 0 --warmup 2 --steps 3
