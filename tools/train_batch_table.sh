#!/bin/bash
# the 6x64 training kernels where they are NOT launch-bound.  For batch 128 / 1024 / 4096: steps/s of the
# step as it ships (graph off: plain launches, side stream on) and, from a rocprofv3 --kernel-trace --stats run with the
# launches in line, the average duration of the convolution / filter-gradient kernels with their algorithmic and issued
# fractions of the dense f16 MFMA peak.  Output: gpurun_out/train_batch_table.json (copied to profiles/ by hand).
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/train_batch_table; rm -rf $OUT; mkdir -p $OUT
for B in 128 1024 4096; do
  S=$((B == 128 ? 300 : (B == 1024 ? 100 : 40)))
  python3 $R/tools/train_native_run.py --steps $S --batch $B > $OUT/run_$B.log 2>&1
  AZX_TRAIN_GRAPH=0 AZX_TRAIN_FORK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$B -- python3 $R/tools/train_native_run.py --steps 12 --batch $B > $OUT/prof_$B.log 2>&1
  cp $(find $OUT/prof_$B -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$B.csv
done
python3 - $OUT <<'P'
import csv, json, re, sys
out = sys.argv[1]
PEAK = 2500e12
table = {"what": "6x64 on 11x11, NativeTrainStep; conv flop per launch = B x 121 x 64 x 64 x 9 x 2 (one layer, one pass)",
         "peak_tflops": 2500.0, "batches": {}}
for B in (128, 1024, 4096):
    m = re.search(r"native step: ([0-9.]+) ms", open("%s/run_%d.log" % (out, B)).read())
    ms = float(m.group(1))
    flop_layer = B * 121 * 64 * 64 * 9 * 2.0
    row = {"ms_per_step": ms, "steps_per_sec": 1e3 / ms, "positions_per_sec": 1e3 / ms * B,
           "algorithmic_tflops_step": 3 * 12 * flop_layer / (ms * 1e-3) / 1e12, "kernels": {}}
    row["frac_of_f16_peak_step"] = row["algorithmic_tflops_step"] * 1e12 / PEAK
    for r in csv.DictReader(open("%s/kernel_stats_%d.csv" % (out, B))):
        name = r["Name"]
        key = ("k_trn_conv<FWD16>" if "k_trn_conv" in name and "Li2E" in name else
               "k_trn_conv<BWD16>" if "k_trn_conv" in name and "Li3E" in name else
               "k_trn_wgrad16" if "k_trn_wgrad16" in name else None)
        if key is None and "k_trn_conv" in name:
            key = name[:40]
        if key is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        row["kernels"][key] = {"avg_us": us, "calls": int(r["Calls"]), "algorithmic_tflops": flop_layer / (us * 1e-6) / 1e12,
                               "frac_of_f16_peak": flop_layer / (us * 1e-6) / PEAK, "issued_frac": 3 * flop_layer / (us * 1e-6) / PEAK}
    table["batches"][str(B)] = row
json.dump(table, open(out + "/../train_batch_table.json", "w"), indent=1)
print(json.dumps(table, indent=1))
P
