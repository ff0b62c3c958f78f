#!/bin/bash
# PMC passes for the tree bench (run on the MI355X box through gpurun).  Counters are collected
# in their own runs (no trace domains besides --kernel-trace), one --pmc group per pass.
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline "${BENCH_ARGS[@]}" > $OUT/$name.json 2> $OUT/$name.err
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2", "fetch", "write"):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (out, name), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:40], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for (kn, cn), (v, n) in sorted(agg.items()):
        if "k_mcts" in kn or "k_tower" in kn:
            print("%-42s %-26s per-dispatch %.6g (n=%d)" % (kn, cn, v / n, n))
PY
