#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/prof_r2.sh into the JSON/CSV files kept under profiles/.

  <out>/summary/kernel_stats.csv        per-kernel totals of the --stats run (copied from rocprofv3's own file)
  <out>/summary/resnet_pmc_traffic.json HBM bytes per k_tower launch (timed launches only), with bench_key
  <out>/summary/tree_pmc_traffic.json   HBM bytes of the nested tree run's timed k_play launch, with bench_key
  <out>/summary/resnet_pmc_counters.json / tree_pmc_counters.json   SQ counters per launch, clock, pipe-busy share

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide loads at 64 bytes,
so reads are doubled (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)


def rows_of(name, suffix):
    rows = []
    for f in glob.glob("%s/%s/**/*%s" % (out, name, suffix), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def bench_line(name):
    try:
        for ln in open(os.path.join(out, name + ".json")):
            if ln.startswith("{"):
                return json.loads(ln)
    except OSError:
        pass
    return None


def is_tower(kn):
    return "k_tower" in kn or "k_conv_wide" in kn


# ---- kernel stats ------------------------------------------------------------------------------
for f in glob.glob("%s/stats/**/*kernel_stats.csv" % out, recursive=True):
    shutil.copy(f, os.path.join(summ, "kernel_stats.csv"))
trace = rows_of("stats", "kernel_trace.csv")
if trace:
    dur = collections.defaultdict(list)
    for r in trace:
        dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    # the timed launches are the last ones of each kernel
    tower = [v for k, v in dur.items() if is_tower(k)]
    info = {}
    for k, v in dur.items():
        if is_tower(k) or "k_heads" in k or "k_play" in k or "k_mcts" in k:
            info[k] = {"launches": len(v), "mean_ms_all": sum(v) / len(v),
                       "mean_ms_timed": sum(v[-steps * 42:]) / len(v[-steps * 42:]) if not "k_play" in k else v[-1]}
    json.dump(info, open(os.path.join(summ, "kernel_durations.json"), "w"), indent=1)

# ---- PMC ---------------------------------------------------------------------------------------
line = bench_line("fetch") or bench_line("stats") or {}
cfg = line.get("config", {})


def per_launch(name, pick):
    """counter -> mean over the picked dispatches"""
    rows = [r for r in rows_of(name, "counter_collection.csv") if pick(r)]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by


def tower_pick(r):
    return is_tower(r["Kernel_Name"]) and float(r["Grid_Size"]) > 100000


def play_pick(r):
    return "k_play" in r["Kernel_Name"]


res = {}
for name in ("fetch", "write", "sq1", "sq2"):
    by = per_launch(name, tower_pick)
    for cn, vals in by.items():
        timed = vals[-steps * 42:] if len(vals) >= steps * 42 else vals
        res[cn] = {"per_launch_mean": sum(timed) / len(timed), "launches": len(timed)}
tree = {}
for name in ("fetch", "write", "sq1", "sq2"):
    by = per_launch(name, play_pick)
    for cn, vals in by.items():
        # the nested tree run's launches come last: settle, warm-up, then ONE launch with all timed moves
        tree[cn] = {"per_launch": vals[-1], "launches_seen": len(vals)}


def traffic(d, key):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"][key], d["WRITE_SIZE"][key]
        return {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                "read_bytes": 2.0 * f * 1024.0, "written_bytes": w * 1024.0,
                "note": "FETCH_SIZE/WRITE_SIZE in KiB per launch, separate --pmc passes; gfx950 tallies the 128-B "
                        "requests of wide loads at 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section)"}
    return None


def args_key(line, nested):
    c = line.get("config", {})
    st = c.get("start", "")
    return line, c


if line:
    games, board, sims, batch = cfg.get("games_per_gpu"), cfg.get("board"), cfg.get("simulations"), cfg.get("search_batch_size")
    import re
    m = re.search(r"0\.\.(\d+) plies, then (\d+) moves", cfg.get("start", ""))
    desync, settle = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    t = traffic(res, "per_launch_mean")
    if t:
        t["bench_key"] = [games, board, sims, batch, 6, 64, steps, warmup, 0.25, desync, settle]
        t["kernel"] = "k_tower_f16x3_s16 (timed launches: last %d)" % res["FETCH_SIZE"]["launches"]
        json.dump(t, open(os.path.join(summ, "resnet_pmc_traffic.json"), "w"), indent=1)
    tt = traffic(tree, "per_launch")
    if tt and "tree" in line:
        tr = line["tree"]
        tt["bench_key"] = [games, board, sims, batch, tr["steps"], tr["warmup"], 0.25, desync, settle]
        tt["kernel"] = "k_play<2> (the nested tree run's timed launch: %d moves)" % tr["steps"]
        json.dump(tt, open(os.path.join(summ, "tree_pmc_traffic.json"), "w"), indent=1)

if res:
    c = {k: v["per_launch_mean"] for k, v in res.items()}
    d = {"kernel": "k_tower_f16x3_s16, per launch (mean of the timed launches)", "counters": c}
    dur = None
    try:
        kd = json.load(open(os.path.join(summ, "kernel_durations.json")))
        dur = [v["mean_ms_timed"] for k, v in kd.items() if is_tower(k)][0]
        d["launch_ms_kernel_trace"] = dur
    except Exception:
        pass
    if "GRBM_GUI_ACTIVE" in c:
        # duration of the same dispatches under the counter pass (kernels serialise under PMC; clock = busy cycles / time)
        ft = [r for r in rows_of("fetch", "kernel_trace.csv") if is_tower(r["Kernel_Name"]) and float(r["Grid_Size_X"]) > 100000]
        ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in ft][-steps * 42:]
        if ms:
            d["launch_ms_fetch_pass"] = sum(ms) / len(ms)
            d["shader_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (sum(ms) / len(ms)) / 1e6
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        s1 = [r for r in rows_of("sq1", "kernel_trace.csv") if is_tower(r["Kernel_Name"]) and float(r["Grid_Size_X"]) > 100000]
        ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in s1][-steps * 42:]
        d["launch_ms_sq1_pass"] = sum(ms) / len(ms) if ms else None
        d["mfma_busy_cycles_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
        d["note"] = ("SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (256 CUs x 4); matrix-pipe busy share = "
                     "that / 1024 / (launch seconds x shader clock); SQ_BUSY_CYCLES is summed per SE/XCD as rocprofv3 reports it")
        if d.get("shader_clock_ghz") and d.get("launch_ms_sq1_pass"):
            d["mfma_pipe_busy_frac"] = d["mfma_busy_cycles_per_simd"] / (d["launch_ms_sq1_pass"] * 1e-3 * d["shader_clock_ghz"] * 1e9)
    json.dump(d, open(os.path.join(summ, "resnet_pmc_counters.json"), "w"), indent=1)
if tree:
    json.dump({"kernel": "k_play<2>, the nested tree run's timed launch", "counters": {k: v["per_launch"] for k, v in tree.items()}},
              open(os.path.join(summ, "tree_pmc_counters.json"), "w"), indent=1)
print(open(os.path.join(summ, "resnet_pmc_counters.json")).read() if res else "no tower counters")
for f in ("resnet_pmc_traffic.json", "tree_pmc_traffic.json", "kernel_durations.json"):
    p = os.path.join(summ, f)
    if os.path.exists(p):
        print(f, open(p).read())
