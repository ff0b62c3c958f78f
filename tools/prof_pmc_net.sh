#!/bin/bash
# PMC passes for the resnet bench's tower kernel (run on the MI355X box through gpurun).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --workload resnet --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tower" in row["Kernel_Name"] and float(row["Grid_Size"]) > 100000:
            k = row["Counter_Name"]; agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%-28s per-dispatch %.6g (n=%d)" % (k, v / n, n))
PY
