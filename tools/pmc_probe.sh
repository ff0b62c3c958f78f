#!/bin/bash
# Diagnostic (tools only): which configuration makes rocprofv3 --pmc fall over on this pool.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_probe; mkdir -p $O
P="rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv"
run() { name=$1; shift; $P -d $O/$name -- python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? segv=$(grep -c SIGSEGV $O/$name.err) $(date +%T) :: $*"; }
run w1 --workload resnet --steps 20 --warmup 5 --no-cpu-baseline --no-replay-exchange
run w2 --workload resnet --steps 20 --warmup 5 --no-cpu-baseline
run w3 --steps 20 --warmup 5 --no-cpu-baseline --api-moves 0 --no-config5
run w4 --steps 20 --warmup 5 --no-cpu-baseline --api-moves 0
