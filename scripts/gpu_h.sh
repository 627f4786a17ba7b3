#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for k in 0 1 2 3 4 5; do
if [ $k = 0 ]; then LIB=""; else LIB=$PWD/variants/libtetris_seg$k.so; fi
TETRIS_MCTS_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --games ${GAMES:-1024} --steps 6 > $OUT/h.seg$k.json 2> $OUT/h.seg$k.err
python - <<PY
import json
d=json.load(open("$OUT/h.seg$k.json"))
print($k, d["last_sim_phase_kcycles"])
PY
done
