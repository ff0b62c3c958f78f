#!/bin/bash
# Diagnostic (tools only): one rocprofv3 --pmc pass over a bench workload for a build of the library; prints the mean
# of every counter per launch of the kernels whose name contains <kernel substring>.
#   tools/pmc.sh <tag> <lib relative to azalea_amd/> <workload: resnet-quick|tree|wide> <kernel substring> <counter> [<counter> ...]
#   e.g. tools/pmc.sh lds libazx_hip.so resnet-quick k_tower SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
#        tools/pmc.sh l2 libazx_hip.so wide k_conv_wide TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# Counters in their own run with --kernel-trace only, the program directly after `--` (gpurun's rules); the wide tower on
# one stream (rocprofv3's counter collection falls over on its second one).
# (replaces pmc_cache / pmc_insts / pmc_lds / l2_hit_wide / pmc_tower_shapes / prof_pmc / prof_pmc_net of rounds 1-3)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; lib=$2; wl=$3; kern=$4; shift 4
OUT=$R/gpurun_out/pmc_$tag; rm -rf $OUT; mkdir -p $OUT
case $wl in
  resnet-quick) ARGS="--workload resnet --steps 3 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange" ;;
  tree)         ARGS="--workload tree --steps 12 --warmup 3 --no-cpu-baseline" ;;
  wide)         export AZX_WIDE_STREAMS=1; ARGS="--workload resnet --board 13 --blocks 19 --chans 256 --games 512 --sims 200 --steps 1 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange" ;;
  *) echo "unknown workload $wl"; exit 2 ;;
esac
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/p -- python3 $R/tools/lib_bench.py $lib $ARGS > $OUT/p.json 2> $OUT/p.err
python3 - $OUT "$kern" <<'PY'
import collections, csv, glob, sys
out, kern = sys.argv[1], sys.argv[2]
agg, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg):
    print("%-36s %18.1f per launch (mean of %d)" % (k, agg[k] / n[k], n[k]))
if not agg:
    print("no launch of a kernel matching '%s' (see %s/p.err)" % (kern, out))
PY
