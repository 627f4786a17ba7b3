#!/bin/bash
# steady-state window at other values of the collector parameters: "<lib|base>:<gc_spec_nodes>:<gc_slice_cycles>" ...
# (lib: a build_variants/<lib>.so of scripts/build_variant.sh)
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-se}; shift
for cfg in "$@"; do
  IFS=: read lib spec slice <<< "$cfg"
  if [ $lib = base ]; then unset TETRIS_MCTS_LIB; else export TETRIS_MCTS_LIB=$PWD/build_variants/$lib.so; fi
  timeout 300 python bench.py --no-cpu-baseline --warmup 75 --steps 20 --steady-steps 0 --gc-spec-nodes $spec --gc-slice-cycles $slice > $OUT/$TAG.${lib}_${spec}_$slice.json 2> $OUT/$TAG.${lib}_${spec}_$slice.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$TAG.${lib}_${spec}_$slice.json"))
    g=d["gc"]
    print("$lib spec=$spec slice=$slice ms/move", round(d["ms_per_step"],1), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "trace", d["mean_trace_len"], "slices/coll", round(g["launches_per_collection"],1), "catchup/move", g["catchup_launches_per_move"], "gc-only", g["collector_only_launches"])
except Exception as e:
    print("$lib spec=$spec slice=$slice failed", e)
PY
done
