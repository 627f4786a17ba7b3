#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -n 3 --maxfail=6 -k "train_dist or real_4096" --durations=5 > $OUT/rp.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 25 $OUT/rp.pytest.log | cut -c1-400
( time timeout 600 python bench.py > $OUT/rp.bench.json 2> $OUT/rp.bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open("gpurun_out/rp.bench.json"))
print({k:d[k] for k in ("value","ms_per_step","mean_trace_len","walk_levels_taken_over_from_the_previous_walk")})
print("roofline", d["roofline"]["kernel"][:25], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "| other", d["roofline_other"]["avg_launch_ms"], d["roofline_other"]["frac"])
print("sum*sims", (d["roofline"]["avg_launch_ms"]+d["roofline_other"]["avg_launch_ms"])*500, "vs", d["ms_per_step"])
print("steady", {k:d["steady_state"][k] for k in ("value","ms_per_step","tree_kernel_ms","value_net_ms")}, d["steady_state"]["gc"])
c=d["cpu_baseline"]; print("cpu", c["kind"], c["cores"], c["value"], c["one_core"], c.get("all_cores"), c.get("python_agent_one_core"), c.get("worker_errors"))
PY
tail -n 3 $OUT/rp.bench.err | cut -c1-300
