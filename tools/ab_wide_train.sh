#!/bin/bash
# Diagnostic (tools only): per-kernel times of the wide training step (19x256 on 13x13, batch 128) for several builds of
# the library, same box, launches in line (AZX_TRAIN_FORK=0): rocprofv3 --kernel-trace --stats, one pass per library.
#   tools/ab_wide_train.sh <spec> [<spec> ...]      spec = [ENV=val[,ENV=val...]@]<lib relative to azalea_amd/>
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AZX_TRAIN_GRAPH=0 AZX_TRAIN_FORK=0
for spec in "$@"; do
  lib=${spec##*@}
  for kv in $(echo "${spec%@*}" | tr ',' ' '); do [ "$spec" != "$lib" ] && export "$kv"; done
  OUT=$R/gpurun_out/ab_wide_train/$(echo $spec | tr '@=,' '___'); rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/lib_run.py $lib $R/tools/train_native_run.py --steps 8 --blocks 19 --chans 256 --board 13 > $OUT/run.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  for kv in $(echo "${spec%@*}" | tr ',' ' '); do [ "$spec" != "$lib" ] && unset "${kv%%=*}"; done
  echo "== $spec: $(grep -o 'native step: [0-9.]* ms' $OUT/run.log)"
  python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_conv_wide_train", "k_tw_wgrad", "k_tw_bnact", "k_tw_bnbwd")):
        print("   %-28s %8.1f us x %s" % (n.split("(")[0].split("<")[0][-28:], float(r["AverageNs"]) / 1e3, r["Calls"]))
P
done
