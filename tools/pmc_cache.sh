#!/bin/bash
# Diagnostic (tools only): vector-L1 (TCP) and L2 (TCC) request counters of the 6x64 tower for a build of the library.
#   tools/pmc_cache.sh <tag> <lib relative to azalea_amd/>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcc_$1
mkdir -p $OUT
ARGS="--workload resnet --steps 3 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p -- python3 $R/tools/lib_bench.py $2 $ARGS > $OUT/p.json 2> $OUT/p.err
python3 - $OUT $1 <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("%s/p/**/*counter_collection.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tower_f16x3_s16" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print(tag, " ".join("%s=%.4g" % (k, v / n) for k, (v, n) in sorted(agg.items())), "(per launch)")
PY
