#!/bin/bash
# Diagnostic (tools only): tree-workload (configs[1]) bench of several builds of the library, one summary line each.
#   tools/ab_tree.sh libazx_hip.so libazx_v0.so ...   (names relative to azalea_amd/; "G:<lib>" = AZX_MCTS_GENERIC=1)
mkdir -p gpurun_out
for spec in "$@"; do
  lib=${spec#G:}; gen=0; [ "$spec" != "$lib" ] && gen=1
  AZX_MCTS_GENERIC=$gen python tools/lib_bench.py $lib --workload tree --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$spec FAILED"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$spec" <<'P'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-22s ms/step %.4f  tree kernel %.4f ms/move  frac %.4f  sims/s %.4g  games/s %.0f  B/sim %.0f" % (
    sys.argv[1], d["ms_per_step"], r["ms_per_move"], r["frac"], d["value"], d["games_per_sec"], r["bytes_per_sim"]))
P
done
