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
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps ${STEPS:-12} --warmup ${WARMUP:-3} --no-cpu-baseline "${BENCH_ARGS[@]}" > $OUT/$name.json 2> $OUT/$name.err
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
        if "k_mcts" in kn or "k_tower" in kn or "k_play" in kn:
            print("%-42s %-26s per-dispatch %.6g (n=%d)" % (kn, cn, v / n, n))
import json
summ = {}
for name in ("fetch", "write"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, name), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_mcts" in r["Kernel_Name"] or "k_tower" in r["Kernel_Name"]
                or "k_play" in r["Kernel_Name"]]
        # skip the warm-up launches: keep the last STEPS dispatches of the dominant kernel -- or, when
        # the moves run in persistent k_play launches, the last dispatch (= all STEPS timed moves)
        import os
        steps = int(os.environ.get("STEPS", "12"))
        persistent = any("k_play" in r["Kernel_Name"] for r in rows)
        if persistent:
            rows = [r for r in rows if "k_play" in r["Kernel_Name"]]
            summ["moves_per_launch"] = steps
        for cn in set(r["Counter_Name"] for r in rows):
            vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == cn][-(1 if persistent else steps):]
            summ[cn] = {"per_launch_mean": sum(vals) / max(1, len(vals)), "launches": len(vals)}
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
    f, w = summ["FETCH_SIZE"]["per_launch_mean"], summ["WRITE_SIZE"]["per_launch_mean"]
    summ["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    summ["note"] = ("FETCH_SIZE/WRITE_SIZE in KiB per launch; on gfx950 FETCH_SIZE tallies 128-B requests of wide "
                    "(16 B/lane) loads at 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section)")
json.dump(summ, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(summ))
PY
