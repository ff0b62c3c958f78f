#!/bin/bash
# Diagnostic (tools only): configs[4]-shape resnet bench under different environment switches.
#   tools/ab_wide.sh "AZX_WIDE_FLIP=0" "AZX_WIDE_FLIP=256" ...
mkdir -p gpurun_out
for spec in "$@"; do
  env $spec python bench.py --workload resnet --board 13 --blocks 19 --chans 256 --sims 800 --games 512 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/abw_tmp.json 2> gpurun_out/abw_tmp.err || { echo "$spec FAILED"; tail -3 gpurun_out/abw_tmp.err; continue; }
  python - "$spec" <<'P'
import json, sys
d = json.loads(open("gpurun_out/abw_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-28s ms/step %.1f  net launch %.2f ms  TFLOP/s %.1f  frac %.4f  issued %.3f  positions/launch %.0f" % (
    sys.argv[1], d["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], r["issued_frac"], r["positions_per_launch"]))
P
done
