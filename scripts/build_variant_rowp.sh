#!/bin/bash
# build_variants/rowp.so: the library with the value net's kernels shaped for CO-RESIDENCY with the tree kernel of another
# sub-batch (the row-P experiment): k_vn_conv asks for 84 KB of LDS per workgroup (so that a CU takes ONE
# workgroup, 1 wave per SIMD at 120 registers, and 76 KB stay free for tree workgroups of 13.3 KB) and its grid is capped at 256
# workgroups.  Measurements only: the product's sources are not touched (the patch is a sed on a copy).
# usage: scripts/build_variant_rowp.sh [NAME=rowp] [LDS_KB=84] [BLOCKS=256]
set -e
NAME=${1:-rowp}; KB=${2:-84}; BL=${3:-256}
cd "$(dirname "$0")/.."
mkdir -p build_variants /tmp/tmv_$NAME
sed -e "s/const int lds = 4 \* WAVE_LDS \* (int)sizeof(float);/const int lds = $KB * 1024;/" \
    -e "s/if (blocks > 256 \* CONV_WG_PER_CU) blocks = 256 \* CONV_WG_PER_CU;/if (blocks > $BL) blocks = $BL;/" \
    tetris_mcts_amd/csrc/valuenet.hip > /tmp/tmv_$NAME/valuenet.hip
cp tetris_mcts_amd/csrc/*.inc /tmp/tmv_$NAME/
grep -c "const int lds = $KB \* 1024;\|if (blocks > $BL) blocks = $BL;" /tmp/tmv_$NAME/valuenet.hip
sed -i 's#"../../include/tetris_mcts_hip.h"#"tetris_mcts_hip.h"#' /tmp/tmv_$NAME/valuenet.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -c /tmp/tmv_$NAME/valuenet.hip -o /tmp/tmv_$NAME/valuenet.o
O=tetris_mcts_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/tree.o $O/search.o $O/core_api.o $O/distnet.o /tmp/tmv_$NAME/valuenet.o -o build_variants/$NAME.so
ls -la build_variants/$NAME.so
