#!/bin/bash
# Builds the product library plus the diagnostic variants (-DAZX_STAMP region timers, =2 wave lifetimes).
set -e
cd "$(dirname "$0")/../azalea_amd/csrc"
make 2>&1 | grep -v "^hipcc" || true
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
for v in 1 2; do
  hipcc $F -DAZX_STAMP=$v -c mcts_kernels.hip -o build/mcts_stamp$v.o
  out=../libazx_stamp.so; [ $v = 2 ] && out=../libazx_stamp2.so
  hipcc --offload-arch=gfx950 -shared -fPIC build/mcts_stamp$v.o build/net_kernels.o build/replay_kernels.o build/azx_capi.o -o $out
done
