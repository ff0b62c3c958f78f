#!/bin/bash
# Diagnostic (tools only): build the library with parts of k_conv_wide_f16x3_s16 ablated (timing only, the
# results are wrong) and time the configs[4]-shape forward: where does a layer's time go?
#   tools/ab_wide_ablate.sh build   (here)          tools/ab_wide_ablate.sh run   (on the MI355X box)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
VARIANTS="${VARIANTS:-1 2 4 8 3 15}"
if [ "$1" = build ]; then
  make -C azalea_amd/csrc 2>&1 | grep -v "^hipcc" || true
  for v in $VARIANTS; do
    hipcc $F -DAZX_WIDE_ABLATE=$v -c azalea_amd/csrc/net_kernels.hip -o azalea_amd/csrc/build/net_ab$v.o &
  done
  wait
  for v in $VARIANTS; do
    hipcc --offload-arch=gfx950 -shared -fPIC azalea_amd/csrc/build/mcts_kernels.o azalea_amd/csrc/build/net_ab$v.o \
      azalea_amd/csrc/build/replay_kernels.o azalea_amd/csrc/build/azx_capi.o -o azalea_amd/libazx_ab$v.so
  done
else
  libs="libazx_hip.so"; for v in $VARIANTS; do libs="$libs libazx_ab$v.so"; done
  bash tools/ab.sh wide $libs
fi
