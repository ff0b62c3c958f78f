#!/bin/bash
# Per-kernel times of the hand-written training step (tools/train_native_run.py) under rocprofv3, the launches in line
# (no graph, no side stream) so that a kernel's duration is its own.  usage: tools/prof_train.sh <tag> [extra args]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_train_$1; shift
rm -rf $OUT; mkdir -p $OUT
AZX_TRAIN_GRAPH=0 AZX_TRAIN_FORK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_native_run.py --steps 30 "$@" > $OUT/run.log 2>&1
grep "native step" $OUT/run.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max([int(r["Calls"]) for r in rows if "k_trn_prep" in r["Name"]] or [1])      # one k_trn_prep per step
tot = 0
for r in rows:
    per_step = float(r["TotalDurationNs"]) / 1e3 / steps
    if "k_trn" in r["Name"] or "rocclr" in r["Name"]:
        tot += per_step
    print("%-58s calls %5s avg %8.1f us  per step %8.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, per_step))
print("sum of the step's kernels: %.1f us" % tot)
P
