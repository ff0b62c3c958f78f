#!/bin/bash
# Diagnostic (tools only): configs[2] bench of several builds of the library (names relative to azalea_amd/).
mkdir -p gpurun_out
for lib in "$@"; do
  python tools/lib_bench.py $lib --workload resnet --no-cpu-baseline > gpurun_out/abr_tmp.json 2> gpurun_out/abr_tmp.err || { echo "$lib FAILED"; tail -3 gpurun_out/abr_tmp.err; continue; }
  python - "$lib" <<'P'
import json, sys
d = json.loads(open("gpurun_out/abr_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-22s ms/step %.1f  net launch pair %.3f ms  TFLOP/s %.1f  frac %.4f  sims/s %.4g" % (
    sys.argv[1], d["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], d["value"]))
P
done
