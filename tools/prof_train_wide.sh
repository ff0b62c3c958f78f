#!/bin/bash
# The wide training step (default 19x256 on 13x13, batch 128) under rocprofv3: per-kernel stats with the launches in
# line, then HBM traffic of its convolution kernels from separate FETCH_SIZE / WRITE_SIZE passes (counters with
# --kernel-trace only, the program directly after `--`).  Writes gpurun_out/prof_train_<tag>/{kernel_stats.csv,summary.json}.
# usage: tools/prof_train_wide.sh <tag> [--blocks 19 --chans 256 --board 13]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
ARGS=${@:---blocks 19 --chans 256 --board 13}
OUT=$R/gpurun_out/prof_train_$tag; rm -rf $OUT; mkdir -p $OUT
export AZX_TRAIN_GRAPH=0 AZX_TRAIN_FORK=0
python3 $R/tools/train_native_run.py --steps 20 $ARGS > $OUT/run_plain.log 2>&1
AZX_TRAIN_FORK=1 python3 $R/tools/train_native_run.py --steps 20 $ARGS > $OUT/run_two_streams.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/train_native_run.py --steps 6 $ARGS > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- python3 $R/tools/train_native_run.py --steps 3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/train_native_run.py --steps 3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 $R/tools/train_native_run.py --steps 3 $ARGS > $OUT/sq1.log 2>&1
python3 - $OUT $ARGS <<'P'
import collections, csv, glob, json, re, sys
out = sys.argv[1]
a = sys.argv[2:]
def arg(name, d):
    return int(a[a.index(name) + 1]) if name in a else d
blocks, chans, board, batch = arg("--blocks", 19), arg("--chans", 256), arg("--board", 13), arg("--batch", 128)
flop = batch * board * board * chans * chans * 9 * 2.0          # one layer, one pass
PEAK = 2500e12
res = {"shape": "%dx%d on %dx%d, batch %d" % (blocks, chans, board, board, batch), "flop_per_conv_launch": flop,
       "algorithmic_bytes_per_conv_launch": {"k_conv_wide_train": "image in (4 B/element) + raw fp32 out: 2 x B x cells x C x 4",
                                             "value": 2.0 * batch * board * board * chans * 4}}
for name in ("run_plain", "run_two_streams"):
    m = re.search(r"native step: ([0-9.]+) ms", open("%s/%s.log" % (out, name)).read())
    res[name + "_ms_per_step"] = float(m.group(1)) if m else None
kern = {}
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    n = r["Name"]
    key = ("k_conv_wide_train_bwd" if "k_conv_wide_train_bwd" in n else "k_conv_wide_train" if "k_conv_wide_train" in n else
           "k_tw_wgrad" if "k_tw_wgrad" in n else "k_tw_bnact" if "k_tw_bnact" in n else "k_tw_bnbwd" if "k_tw_bnbwd" in n else
           "k_trn_update" if "k_trn_update" in n else None)
    if key is None:
        continue
    us = float(r["AverageNs"]) / 1e3
    d = {"avg_us": us, "calls": int(r["Calls"])}
    if key.startswith("k_conv") or key == "k_tw_wgrad":
        d.update({"algorithmic_tflops": flop / (us * 1e-6) / 1e12, "frac_of_f16_peak": flop / (us * 1e-6) / PEAK,
                  "issued_frac": 3 * flop / (us * 1e-6) / PEAK})
    kern[key] = d
def counters(name):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, name), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = ("k_conv_wide_train_bwd" if "k_conv_wide_train_bwd" in n else "k_conv_wide_train" if "k_conv_wide_train" in n else
                   "k_tw_wgrad" if "k_tw_wgrad" in n else None)
            if key:
                by[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by
for name in ("fetch", "write", "sq1"):
    for key, cs in counters(name).items():
        for cn, v in cs.items():
            kern.setdefault(key, {})[cn] = sum(v) / len(v)
for key, d in kern.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0      # gfx950: reads doubled (MI355X_MICROARCH.md)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "avg_us" in d and "GRBM_GUI_ACTIVE" in d:
        clk = d["GRBM_GUI_ACTIVE"] / 8.0 / (d["avg_us"] * 1e-6)
        d["shader_clock_ghz_fetch_pass"] = clk / 1e9
        d["mfma_pipe_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["avg_us"] * 1e-6 * clk)
res["kernels"] = kern
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
