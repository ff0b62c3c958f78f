#!/bin/bash
# Diagnostic (tools only): configs[2] resnet bench (lock-step start, 3 moves) of several builds of the library:
# tower / heads / tree kernel times per leaf batch from the bench line.
mkdir -p gpurun_out
for lib in "$@"; do
  python tools/lib_bench.py $lib --workload resnet --steps 3 --warmup 1 --desync 0 --no-cpu-baseline --no-replay-exchange > gpurun_out/abr_tmp.json 2> gpurun_out/abr_tmp.err || { echo "$lib FAILED"; tail -3 gpurun_out/abr_tmp.err; continue; }
  python - "$lib" <<'P'
import json, sys
d = json.loads(open("gpurun_out/abr_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-22s ms/step %.2f  net launch pair %.3f ms  TFLOP/s %.1f  frac %.4f" % (sys.argv[1], d["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"]))
P
done
