#!/bin/bash
# the convolution kernel: bit-exactness against the oracle, the value net inside the search, the headline bench
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-cv}
timeout 600 python -m pytest tests/test_gpu_valuenet.py tests/test_gpu_tree.py -m gpu -q -n 4 --maxfail=4 -k "valuenet or value_net_in_the_loop or reference_golden_runs or online_training_loop" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 $OUT/$TAG.pytest.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --steady-steps 0 > $OUT/$TAG.fresh.json 2> $OUT/$TAG.fresh.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/$TAG.fresh.json"))
    print("fresh ms/move", round(d["ms_per_step"],2), "value", round(d["value"]/1e6,3), "nn", round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "trace", d["mean_trace_len"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/$TAG.fresh.err").read()[-1500:])
PY
