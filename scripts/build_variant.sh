#!/bin/bash
# build a variant of the HIP library with extra -D flags: scripts/build_variant.sh NAME -DFOO ...  -> variants/libtetris_NAME.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p variants/_obj_$NAME
for f in tree core_api valuenet search; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c tetris_mcts_amd/csrc/$f.hip -o variants/_obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/_obj_$NAME/*.o -o variants/libtetris_$NAME.so
echo built variants/libtetris_$NAME.so
