#!/bin/bash
# Diagnostic (tools only): configs[4]-shape resnet bench of several builds of the library.
mkdir -p gpurun_out
for lib in "$@"; do
  python tools/lib_bench.py $lib --workload resnet --board 13 --blocks 19 --chans 256 --sims 800 --games 512 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/abw_tmp.json 2> gpurun_out/abw_tmp.err || { echo "$lib FAILED"; tail -3 gpurun_out/abw_tmp.err; continue; }
  python - "$lib" <<'P'
import json, sys
d = json.loads(open("gpurun_out/abw_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-22s net launch %.2f ms  TFLOP/s %.1f  issued %.3f" % (sys.argv[1], r["avg_launch_ms"], r["achieved"], r["issued_frac"]))
P
done
