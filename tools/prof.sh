#!/bin/bash
# Profile set of the command the driver runs (python3 bench.py --steps 20 --warmup 5), on the MI355X
# box through gpurun.  Kernel trace + stats in one run; PMC counters in their own runs (one --pmc group per
# pass, --kernel-trace only, the program directly after `--`).  The line's three GPU legs are all in the run:
# the configs[2] headline (k_tower_f16x3_s16), the nested configs[1] tree run (k_play<2>) and the nested
# configs[4]-shape leg (k_conv_wide_f16x3_s16); the product-surface leg is skipped (--api-moves 0: the same kernels
# as the headline).  Summaries land in gpurun_out/<out>/summary/ (tools/prof_summarise.py) and are copied into
# profiles/ by hand.
# usage: tools/prof.sh <outdir under gpurun_out> [passes...]   passes: stats fetch write sq1 sq2 (default all)
set -u
cd /tmp
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
PASSES=${@:-stats fetch write sq1 sq2}
# The timing pass is the driver's command itself.  rocprofv3's counter collection, however, falls over on this pool
# (SIGSEGV in the dispatch intercept / its worker thread, all output lost) (a) when an engine launches on a second
# HIP stream, which the wide tower does (two halves of a batch on two streams), and (b) somewhere
# past ~10 k dispatches in one process -- the configs[4]-shape leg alone is 12.8 k.  So every counter pass is two
# runs: the headline + nested tree run (--no-config5), and the configs[4]-shape leg on its own with the wide tower
# on ONE stream and 200 instead of 800 sims per move (21 instead of 81 leaf batches per move: the same kernels
# on the same 512-game batches, so the same bytes and cycles per forward, which is what the summary reports).
export TMPDIR=/tmp
# --no-train-step: the training-step leg runs each mode in a CHILD process, which would inherit the profiler's
# preload and write into the same -d directory (tools/prof_train.sh profiles the training step on its own)
ARGS="--steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --api-moves 0 --no-train-step ${EXTRA:-}"
C5="--workload resnet --board 13 --blocks 19 --chans 256 --sims 200 --games ${C5GAMES:-512} --steps 1 --warmup 0 --no-cpu-baseline --no-replay-exchange --no-train-step"
pmc() {   # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py $ARGS --no-config5 > $OUT/$name.json 2> $OUT/$name.err
  AZX_WIDE_STREAMS=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${name}_c5 -- python3 $R/bench.py $C5 > $OUT/${name}_c5.json 2> $OUT/${name}_c5.err
  echo "$name: segv $(grep -c SIGSEGV $OUT/$name.err) / $(grep -c SIGSEGV $OUT/${name}_c5.err)"
}
for p in $PASSES; do
  case $p in
    stats) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.err ;;
    fetch) pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE ;;
    write) pmc write WRITE_SIZE ;;
    sq1)   pmc sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU ;;
    sq2)   pmc sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR ;;
  esac
done
python3 $R/tools/prof_summarise.py $OUT ${STEPS:-20} ${WARMUP:-5}
