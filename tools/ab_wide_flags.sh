#!/bin/bash
# Diagnostic (tools only): build the library with different -D switches for net_kernels.hip and time the
# configs[4]-shape forward.   tools/ab_wide_flags.sh build "name1:-DX=1" "name2:-DX=2 -DY=3" ...   (here)
#                             tools/ab_wide_flags.sh run name1 name2 ...                           (MI355X box)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
mode=$1; shift
if [ "$mode" = build ]; then
  make -C azalea_amd/csrc 2>&1 | grep -v "^hipcc" || true
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( hipcc $F $flags -c azalea_amd/csrc/net_kernels.hip -o azalea_amd/csrc/build/net_$name.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC azalea_amd/csrc/build/mcts_kernels.o azalea_amd/csrc/build/net_$name.o \
        azalea_amd/csrc/build/replay_kernels.o azalea_amd/csrc/build/azx_capi.o -o azalea_amd/libazx_$name.so ) &
  done
  wait
else
  libs="libazx_hip.so"; for n in "$@"; do libs="$libs libazx_$n.so"; done
  bash tools/ab.sh wide $libs
fi
