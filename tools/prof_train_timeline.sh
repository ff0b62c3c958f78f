#!/bin/bash
# The training step as it ships (stream launches, weight gradients on their own stream; AZX_TRAIN_GRAPH=1 for the captured graph) under rocprofv3's kernel trace,
# then the timeline of its last step (tools/train_timeline.py).  usage: tools/prof_train_timeline.sh <tag>
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_traintl_$1; shift
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/train_native_run.py --steps 20 "$@" > $OUT/run.log 2>&1
grep "native step" $OUT/run.log
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/tools/train_timeline.py $f > $OUT/timeline.txt
cat $OUT/timeline.txt
