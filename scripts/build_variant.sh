#!/bin/bash
# build_variants/<name>.so: the library with other values of tree.hip's tuning constants (measurements only; runs where hipcc is)
# usage: scripts/build_variant.sh NAME 'sed-expression' ['sed-expression' ...]      then: TETRIS_MCTS_LIB=build_variants/NAME.so python bench.py ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants /tmp/tmv_$NAME
cp tetris_mcts_amd/csrc/tree.hip /tmp/tmv_$NAME/tree.hip
for e in "$@"; do sed -i -e "$e" /tmp/tmv_$NAME/tree.hip; done
if cmp -s tetris_mcts_amd/csrc/tree.hip /tmp/tmv_$NAME/tree.hip; then echo "build_variant $NAME: no expression matched"; exit 1; fi
sed -i 's#"../../include/tetris_mcts_hip.h"#"'$PWD'/include/tetris_mcts_hip.h"#' /tmp/tmv_$NAME/tree.hip
cp tetris_mcts_amd/csrc/*.h tetris_mcts_amd/csrc/*.inc /tmp/tmv_$NAME/ 2>/dev/null || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c /tmp/tmv_$NAME/tree.hip -o /tmp/tmv_$NAME/tree.o
O=tetris_mcts_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/tmv_$NAME/tree.o $O/search.o $O/core_api.o $O/valuenet.o $O/distnet.o $O/yogi.o -o build_variants/$NAME.so
echo "build_variants/$NAME.so"
