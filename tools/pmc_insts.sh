#!/bin/bash
# Diagnostic (tools only): dynamic instruction counts of k_mcts for a build of the library.
#   tools/pmc_insts.sh <tag> <lib relative to azalea_amd/> [AZX_MCTS_GENERIC value]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmci_$1
mkdir -p $OUT
export AZX_MCTS_GENERIC=${3:-0}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/sq1 -- python3 $R/tools/lib_bench.py $2 --workload tree --steps 12 --warmup 3 --no-cpu-baseline > $OUT/sq1.json 2> $OUT/sq1.err
python3 - $OUT $1 <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("%s/sq1/**/*counter_collection.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_mcts" in row["Kernel_Name"] or "k_play" in row["Kernel_Name"]:
            k = row["Counter_Name"]; agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
sims = 4096 * 410.0   # per move; k_play launches cover several moves (warm-up 3 + timed 12 here)
print(tag, " ".join("%s/sim=%.1f" % (k.replace("SQ_INSTS_", ""), v / n / sims) for k, (v, n) in sorted(agg.items()) if k != "SQ_WAVES"))
PY
