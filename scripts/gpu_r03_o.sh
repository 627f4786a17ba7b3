#!/bin/bash
# steady-state windows of the other legs: the online leg (tuples harvested and exchanged) and ValueSimLP
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --online --warmup 75 --steps 20 --steady-steps 0 > $OUT/ev.steady_online.json 2> $OUT/ev.steady_online.err; echo "online rc=$?"
timeout 900 python bench.py --agent ValueSimLP --no-cpu-baseline --warmup 75 --steps 20 --steady-steps 0 > $OUT/ev.steady_lp.json 2> $OUT/ev.steady_lp.err; echo "lp rc=$?"
python - <<PY
import json
for n in ("online","lp"):
    try:
        d=json.load(open("$OUT/ev.steady_%s.json"%n)); g=d["gc"]
        print(n, "ms/move", round(d["ms_per_step"],1), "value", round(d["value"]/1e6,3), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "slices/coll", g["launches_per_collection"], "catchup/move", g["catchup_launches_per_move"], "collections", g["collections"], "dropped", g["dropped_tuples"], d.get("exchange"))
    except Exception as e: print(n, "failed", e)
PY
