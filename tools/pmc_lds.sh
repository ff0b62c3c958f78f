#!/bin/bash
# Diagnostic (tools only): LDS counters of the network kernels for a build of the library.
#   tools/pmc_lds.sh <tag> <lib relative to azalea_amd/> [resnet|config5]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcl_$1
mkdir -p $OUT
if [ "${3:-resnet}" = config5 ]; then
  export AZX_WIDE_STREAMS=1
  ARGS="--workload resnet --board 13 --blocks 19 --chans 256 --games 512 --sims 200 --steps 1 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange"
else
  ARGS="--workload resnet --steps 3 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange"
fi
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p -- python3 $R/tools/lib_bench.py $2 $ARGS > $OUT/p.json 2> $OUT/p.err
python3 - $OUT $1 <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("%s/p/**/*counter_collection.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if "tower" in name or "conv_wide" in name:
            a = agg[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for name, c in agg.items():
    n = c["SQ_INSTS_LDS"][0]
    print(tag, name, "launches=%d" % c["SQ_INSTS_LDS"][1], " ".join("%s/inst=%.2f" % (k, v / n) for k, (v, _) in sorted(c.items()) if k != "SQ_INSTS_LDS"))
PY
