#!/bin/bash
# Kernel timeline of the play-ahead loop (tools/bench_train_loop.py --one overlapped): are the self-play tower and the
# training step's kernels RESIDENT TOGETHER?  rocprofv3 --kernel-trace (no counters), then tools/overlap_summary.py over the
# trace: per kernel family, the share of its device time during which a kernel of the other side was running too.
#   tools/prof_overlap.sh <tag> [bench_train_loop args]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
OUT=$R/gpurun_out/prof_overlap_$tag; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/bench_train_loop.py --one overlapped --steps ${STEPS:-600} "$@" > $OUT/run.json 2> $OUT/run.err
python3 $R/tools/overlap_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/summary.txt
cat $OUT/summary.txt
