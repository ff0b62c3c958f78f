#!/bin/bash
# Diagnostic (tools only): same-box A/B of library builds and / or environment switches, one summary line per spec.
#   tools/ab.sh <workload> <spec> [<spec> ...]
#     workload  resnet (configs[2], full steady-state bench) | resnet-quick (lock-step start, 3 moves) | tree (configs[1]) |
#               wide (configs[4] shape: 13x13, 19x256, 512 games, one move) | train (tools/train_native_run.py)
#     spec      [ENV=val[,ENV=val...]@]<lib relative to azalea_amd/>      e.g.  libazx_hip.so   AZX_HEADS=valu@libazx_hip.so
# (replaces the round 1-3 one-off scripts ab_heads / ab_resnet / ab_resnet_lib / ab_tree / ab_wide / ab_wide_lib)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
wl=$1; shift
case $wl in
  resnet)       ARGS="--workload resnet --no-cpu-baseline" ;;
  resnet-quick) ARGS="--workload resnet --steps 3 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange" ;;
  tree)         ARGS="--workload tree --no-cpu-baseline" ;;
  wide)         ARGS="--workload resnet --board 13 --blocks 19 --chans 256 --sims 800 --games 512 --steps 1 --warmup 1 --no-cpu-baseline --no-replay-exchange" ;;
  train)        ARGS="" ;;
  *) echo "unknown workload $wl"; exit 2 ;;
esac
for spec in "$@"; do
  lib=${spec##*@}; envs=""
  [ "$spec" != "$lib" ] && envs=$(echo "${spec%@*}" | tr ',' ' ')
  if [ $wl = train ]; then
    echo "$spec: $(env $envs python tools/lib_run.py $lib tools/train_native_run.py --steps 300 2>/dev/null | head -1)"
    continue
  fi
  env $envs python tools/lib_bench.py $lib $ARGS > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$spec FAILED"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$spec" <<'P'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
extra = "  tree %.4f ms/move" % r["ms_per_move"] if "ms_per_move" in r else "  issued %.3f" % r.get("issued_frac", 0)
print("%-40s ms/step %8.2f  dominant launch %.3f ms  %s %.1f  frac %.4f%s  sims/s %.4g" % (
    sys.argv[1], d["ms_per_step"], r["avg_launch_ms"], r["unit"], r["achieved"], r["frac"], extra, d["value"]))
P
done
