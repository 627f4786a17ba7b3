#!/bin/bash
# build_variants/<name>.so: the library with one source file edited by sed expressions (measurements only; runs where hipcc is)
# usage: scripts/build_variant.sh NAME FILE 'sed-expression' ['sed-expression' ...]     FILE: tree.hip | valuenet.hip | valuenet_conv.inc | distnet.hip
# then:  TETRIS_MCTS_LIB=$PWD/build_variants/NAME.so python bench.py ...
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/.."
D=/tmp/tmv_$NAME; rm -rf $D; mkdir -p build_variants $D
cp tetris_mcts_amd/csrc/*.hip tetris_mcts_amd/csrc/*.h tetris_mcts_amd/csrc/*.inc $D/
for e in "$@"; do sed -i -e "$e" $D/$FILE; done
if cmp -s tetris_mcts_amd/csrc/$FILE $D/$FILE; then echo "build_variant $NAME: no expression matched in $FILE"; exit 1; fi
sed -i 's#"../../include/tetris_mcts_hip.h"#"'$PWD'/include/tetris_mcts_hip.h"#' $D/*.hip
SRC=$FILE; [ $FILE = valuenet_conv.inc ] && SRC=valuenet.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c $D/$SRC -o $D/variant.o
O=tetris_mcts_amd/csrc/_obj; OBJS=""
for s in tree core_api valuenet distnet search yogi; do if [ $s.hip = $SRC ]; then OBJS="$OBJS $D/variant.o"; else OBJS="$OBJS $O/$s.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o build_variants/$NAME.so
echo "build_variants/$NAME.so"
