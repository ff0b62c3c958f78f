#!/bin/bash
# Round-3 profile set of the command the driver runs (python3 bench.py --steps 20 --warmup 5), on the MI355X
# box through gpurun.  Kernel trace + stats in one run; PMC counters in their own runs (one --pmc group per
# pass, --kernel-trace only, the program directly after `--`).  The line's three GPU legs are all in the run:
# the configs[2] headline (k_tower_f16x3_s16), the nested configs[1] tree run (k_play<2>) and the nested
# configs[4]-shape leg (k_conv_wide_f16x3_s16); the product-surface leg is skipped (--api-moves 0: the same kernels
# as the headline).  Summaries land in gpurun_out/<out>/summary/ (tools/prof_r3_summarise.py) and are copied into
# profiles/ by hand.
# usage: tools/prof_r3.sh <outdir under gpurun_out> [passes...]   passes: stats fetch write sq1 sq2 (default all)
set -u
cd /tmp
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
PASSES=${@:-stats fetch write sq1 sq2}
# rocprofv3's counter collection falls over (SIGSEGV inside the dispatch intercept) when an engine launches on a
# second HIP stream, which the wide tower does (two halves of a batch on two streams): the counter
# passes run it on one stream -- same kernels, same bytes and cycles per forward; the timing pass keeps two.
PMCENV="AZX_WIDE_STREAMS=1"
ARGS="--steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --api-moves 0 ${EXTRA:-}"
for p in $PASSES; do
  case $p in
    stats) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.err ;;
    fetch) export $PMCENV; rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err ;;
    write) export $PMCENV; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err ;;
    sq1)   export $PMCENV; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > $OUT/sq1.json 2> $OUT/sq1.err ;;
    sq2)   export $PMCENV; rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err ;;
  esac
done
python3 $R/tools/prof_r3_summarise.py $OUT ${STEPS:-20} ${WARMUP:-5}
