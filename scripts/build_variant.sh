#!/bin/bash
# build_variants/<name>.so: the library with other values of the collector tuning constants (measurements only)
# usage: scripts/build_variant.sh NAME GC_COST_MAX GC_BLOCKS_MAX [COST_NODES COST_OBS]
set -e
NAME=$1; COST=$2; BLOCKS=$3; CN=${4:-5}; CO=${5:-5}
cd "$(dirname "$0")/.."
mkdir -p build_variants /tmp/tmv_$NAME
sed -e "s/^constexpr int GC_COST_MAX = [0-9]*;/constexpr int GC_COST_MAX = $COST;/" \
    -e "s/^constexpr int GC_BLOCKS_MAX = [0-9]*;/constexpr int GC_BLOCKS_MAX = $BLOCKS;/" \
    -e "s/^static_assert(3 \* GC_BLOCKS_MAX <= TM_GC_PART_DW/static_assert(3 * (GC_BLOCKS_MAX \/ 2) <= TM_GC_PART_DW/" \
    -e "s/const int cost = step == GCP_WRITE ? 2 : (step == GCP_NODES || step == GCP_OBS) ? 5 : 1;/const int cost = step == GCP_WRITE ? 2 : step == GCP_NODES ? $CN : step == GCP_OBS ? $CO : 1;/" \
    tetris_mcts_amd/csrc/tree.hip > /tmp/tmv_$NAME/tree.hip
cp tetris_mcts_amd/csrc/*.h tetris_mcts_amd/csrc/*.inc /tmp/tmv_$NAME/ 2>/dev/null || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Itetris_mcts_amd/csrc -c /tmp/tmv_$NAME/tree.hip -o /tmp/tmv_$NAME/tree.o
O=tetris_mcts_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/tmv_$NAME/tree.o $O/search.o $O/core_api.o $O/valuenet.o -o build_variants/$NAME.so
grep -c "GC_COST_MAX = $COST;\|GC_BLOCKS_MAX = $BLOCKS;\|step == GCP_NODES ? $CN : step == GCP_OBS ? $CO" /tmp/tmv_$NAME/tree.hip
