#!/bin/bash
# Diagnostic (tools only): configs[2] with the MFMA heads and with the scalar-FMA heads (AZX_HEADS=valu), alternating.
mkdir -p gpurun_out
for h in mfma valu mfma valu; do  # (libazx_hip.so)
  AZX_HEADS=$h python bench.py --workload resnet --steps 6 --warmup 2 --no-cpu-baseline --no-replay-exchange > gpurun_out/abh_tmp.json 2> gpurun_out/abh_tmp.err || { echo "$h FAILED"; tail -3 gpurun_out/abh_tmp.err; continue; }
  python - "$h" <<'P'
import json, sys
d = json.loads(open("gpurun_out/abh_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("heads=%-5s ms/step %.2f  net launch pair %.3f ms  frac %.4f  sims/s %.4g" % (sys.argv[1], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["value"]))
P
done
