#!/bin/bash
# r03: GC parity tests, default bench, steady-state bench (moves 76-95)
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-rd}
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/$TAG.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/$TAG.smoke.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -n 6 --maxfail=6 -k "with_gc or beyond_128 or replay_harvest or pool_exhaustion or single_calls or online_training_loop or sampled_seeds" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 15 $OUT/$TAG.pytest.log | cut -c1-300
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], "ms/move", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "err", d["error_games"], "len", round(d["mean_trace_len"],2), d["max_trace_len"], "exp/s", round(d["value"]), "sims/s", round(d["sims_per_sec"]), d["gc"], {k: round(v,1) for k,v in d["last_sim_phase_kcycles"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout 200 python bench.py --no-cpu-baseline > $OUT/$TAG.bench.json 2> $OUT/$TAG.bench.err; echo "bench rc=$?"; summ $OUT/$TAG.bench.json; tail -n 2 $OUT/$TAG.bench.err | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --warmup 75 --steps 20 > $OUT/$TAG.steady.json 2> $OUT/$TAG.steady.err; echo "steady rc=$?"; summ $OUT/$TAG.steady.json; tail -n 2 $OUT/$TAG.steady.err | cut -c1-300
