#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/prof.sh into the JSON/CSV files kept under profiles/.

  <out>/summary/kernel_stats.csv          per-kernel totals of the --stats run (rocprofv3's own file)
  <out>/summary/kernel_durations.json     mean duration of the timed launches of the tower / wide-conv / heads / tree kernels
  <out>/summary/resnet_pmc_traffic.json   HBM bytes per k_tower_f16x3_s16 launch (timed launches), with bench_key
  <out>/summary/tree_pmc_traffic.json     HBM bytes of the nested tree run's timed k_play<2> launch, with bench_key
  <out>/summary/config5_pmc_traffic.json  HBM bytes per leaf-batch forward of the nested configs[4]-shape leg (stem + 38 layer
                                          launches, mean over all of the leg's forwards), with bench_key
  <out>/summary/{resnet,tree,config5}_pmc_counters.json   SQ counters, shader clock, matrix-pipe busy share

Kernels are told apart by NAME (the three legs use different kernels: k_tower_f16x3_s16 / k_play<2> /
k_conv_wide_f16x3_s16 + k_stem_wide_f16x3; the configs[4]-shape leg's pool set-up runs k_play<3>).
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide loads at 64 bytes,
so reads are doubled (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

out, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)
NSIMD = 1024.0


def rows_of(name, suffix):
    rows = []
    for f in sorted(glob.glob("%s/%s/**/*%s" % (out, name, suffix), recursive=True)):
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
    return "k_tower_f16x3_s16" in kn


def is_wide(kn):
    return "k_conv_wide_f16x3" in kn


def is_wide_any(kn):
    return is_wide(kn) or "k_stem_wide" in kn


def is_play2(kn):
    return "k_playILi2" in kn or "k_play<2>" in kn


def dur_ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def src_sha(ln):
    """`src=<digest>` of the bench line's kernel description: the kernel sources the profiled library was built from."""
    for part in ((ln or {}).get("kernels") or "").split(";"):
        if part.strip().startswith("src="):
            return part.strip()[4:]
    return None


line = bench_line("stats") or bench_line("fetch") or {}
SHA = src_sha(bench_line("fetch") or line)          # the counters' own run
evals_per_move = 42          # 41 select batches + the root evaluation (11x11, 400 sims)

# ---- kernel stats ------------------------------------------------------------------------------
for f in glob.glob("%s/stats/**/*kernel_stats.csv" % out, recursive=True):
    shutil.copy(f, os.path.join(summ, "kernel_stats.csv"))
trace = rows_of("stats", "kernel_trace.csv")
if trace:
    dur = collections.defaultdict(list)
    for r in trace:
        dur[r["Kernel_Name"][:70]].append(dur_ms(r))
    info = {}
    for k, v in dur.items():
        if is_tower(k) or "k_heads" in k:
            n = steps * evals_per_move
            info[k] = {"launches": len(v), "mean_ms_all": sum(v) / len(v)}
            if is_tower(k):
                info[k]["mean_ms_timed"] = sum(v[-n:]) / len(v[-n:])
        elif is_wide_any(k):
            info[k] = {"launches": len(v), "mean_ms_all": sum(v) / len(v), "total_ms": sum(v)}
        elif "k_play" in k or "k_mcts" in k:
            info[k] = {"launches": len(v), "mean_ms_all": sum(v) / len(v), "last_ms": v[-1]}
    json.dump(info, open(os.path.join(summ, "kernel_durations.json"), "w"), indent=1)


def counters(name, pick):
    by = collections.defaultdict(list)
    for r in rows_of(name, "counter_collection.csv"):
        if pick(r["Kernel_Name"]):
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by


def traffic(f, w):
    return {"src_sha": SHA, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
            "read_bytes": 2.0 * f * 1024.0, "written_bytes": w * 1024.0,
            "note": "FETCH_SIZE/WRITE_SIZE in KiB, separate --pmc passes; gfx950 tallies the 128-B requests of wide "
                    "loads at 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section)"}


cfg = line.get("config", {})
m = re.search(r"0\.\.(\d+) plies, then (\d+) moves", cfg.get("start", ""))
desync, settle = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
games, board, sims, batch = cfg.get("games_per_gpu"), cfg.get("board"), cfg.get("simulations"), cfg.get("search_batch_size")
ntimed = steps * evals_per_move

# ---- headline: k_tower_f16x3_s16 ------------------------------------------------------------------
res = {}
for name in ("fetch", "write", "sq1", "sq2"):
    for cn, vals in counters(name, is_tower).items():
        timed = vals[-ntimed:]
        res[cn] = sum(timed) / len(timed)
heads = {}
for name in ("fetch", "write"):
    for cn, vals in counters(name, lambda kn: "k_heads" in kn).items():
        timed = vals[-ntimed:]
        heads[cn] = sum(timed) / len(timed)
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    t = traffic(res["FETCH_SIZE"], res["WRITE_SIZE"])
    t["bench_key"] = [games, board, sims, batch, 6, 64, steps, warmup, 0.25, desync, settle]
    t["kernel"] = "k_tower_f16x3_s16 (mean of the %d timed launches; the k_heads launch behind each is heads_hbm_bytes_per_launch)" % ntimed
    if "FETCH_SIZE" in heads and "WRITE_SIZE" in heads:
        t["heads_hbm_bytes_per_launch"] = (2.0 * heads["FETCH_SIZE"] + heads["WRITE_SIZE"]) * 1024.0
        t["heads_FETCH_SIZE_KiB"], t["heads_WRITE_SIZE_KiB"] = heads["FETCH_SIZE"], heads["WRITE_SIZE"]
    t["algorithmic_bytes_per_launch"] = "boards in: positions x 192 B; head planes out: positions x 6 x 121 x 4 B (~119 MB at 40 960 positions)"
    json.dump(t, open(os.path.join(summ, "resnet_pmc_traffic.json"), "w"), indent=1)


def pass_ms(name, pick, last=None):
    v = [dur_ms(r) for r in rows_of(name, "kernel_trace.csv") if pick(r["Kernel_Name"])]
    v = v[-last:] if last else v
    return sum(v) / len(v) if v else None


def pipe(d, c, fetch_ms, sq1_ms):
    if "GRBM_GUI_ACTIVE" in c and fetch_ms:
        d["launch_ms_fetch_pass"] = fetch_ms
        d["shader_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / fetch_ms / 1e6
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and sq1_ms:
        d["launch_ms_sq1_pass"] = sq1_ms
        d["mfma_busy_cycles_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / NSIMD
        if d.get("shader_clock_ghz"):
            d["mfma_pipe_busy_frac"] = d["mfma_busy_cycles_per_simd"] / (sq1_ms * 1e-3 * d["shader_clock_ghz"] * 1e9)
    d["note"] = ("SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (256 CUs x 4); matrix-pipe busy share = that / 1024 / "
                 "(launch seconds x shader clock); shader clock = GRBM_GUI_ACTIVE / 8 XCDs / launch time of the same pass")


if res:
    d = {"kernel": "k_tower_f16x3_s16, per launch (mean of the timed launches)", "counters": res}
    try:
        kd = json.load(open(os.path.join(summ, "kernel_durations.json")))
        d["launch_ms_kernel_trace"] = [v["mean_ms_timed"] for k, v in kd.items() if is_tower(k)][0]
    except Exception:
        pass
    pipe(d, res, pass_ms("fetch", is_tower, ntimed), pass_ms("sq1", is_tower, ntimed))
    json.dump(d, open(os.path.join(summ, "resnet_pmc_counters.json"), "w"), indent=1)

# ---- nested tree run: the last k_play<2> launch ---------------------------------------------------
tree = {}
for name in ("fetch", "write", "sq1", "sq2"):
    for cn, vals in counters(name, is_play2).items():
        tree[cn] = vals[-1]
if "FETCH_SIZE" in tree and "WRITE_SIZE" in tree and "tree" in line:
    tr = line["tree"]
    tt = traffic(tree["FETCH_SIZE"], tree["WRITE_SIZE"])
    tt["bench_key"] = [games, board, sims, batch, tr["steps"], tr["warmup"], 0.25, desync, settle]
    tt["kernel"] = "k_play<2> (the nested tree run's timed launch: %d moves)" % tr["steps"]
    json.dump(tt, open(os.path.join(summ, "tree_pmc_traffic.json"), "w"), indent=1)
if tree:
    json.dump({"kernel": "k_play<2>, the nested tree run's timed launch", "counters": tree},
              open(os.path.join(summ, "tree_pmc_counters.json"), "w"), indent=1)

# ---- configs[4]-shape leg: k_stem_wide + 38 x k_conv_wide per leaf-batch forward ------------------
# counters from the leg's own runs (<pass>_c5: 200 sims per move, one stream); timing from the driver's command
c5 = line.get("config5")
c5line = bench_line("fetch_c5") or bench_line("write_c5") or {}
if c5 or c5line:
    sums, n_fwd = {}, {}
    for name in ("fetch_c5", "write_c5", "sq1_c5", "sq2_c5"):
        rows = [r for r in rows_of(name, "counter_collection.csv") if is_wide_any(r["Kernel_Name"])]
        fw = len({r["Dispatch_Id"] for r in rows if "k_stem_wide" in r["Kernel_Name"]})
        for r in rows:
            cn = r["Counter_Name"]
            sums[cn] = sums.get(cn, 0.0) + float(r["Counter_Value"])
            n_fwd[cn] = fw
    per_fwd = {cn: v / max(1, n_fwd[cn]) for cn, v in sums.items()}
    heads5 = {}
    for name in ("fetch_c5", "write_c5"):
        for cn, vals in counters(name, lambda kn: "k_heads" in kn).items():
            heads5[cn] = sum(vals) / len(vals)
    cc = (c5 or c5line)["config"]
    key5 = [cc["games_per_gpu"], 13, cc["search_batch_size"], 19, 256, "per forward"]
    pos = (c5line.get("roofline") or (c5 or {}).get("roofline") or {}).get("positions_per_launch")
    if "FETCH_SIZE" in per_fwd and "WRITE_SIZE" in per_fwd:
        t5 = traffic(per_fwd["FETCH_SIZE"], per_fwd["WRITE_SIZE"])
        t5["bench_key"] = key5
        t5["kernel"] = ("k_stem_wide_f16x3 + 38 x k_conv_wide_f16x3_s16 per leaf-batch forward: mean over %d forwards of the "
                        "configs[4]-shape leg run on its own under the counters (200 sims per move, wide tower on one "
                        "stream: rocprofv3 --pmc falls over on the second stream and past ~10 k dispatches; same kernels, "
                        "same 512-game batches; k_heads not included)" % n_fwd["FETCH_SIZE"])
        t5["positions_per_forward"] = pos
        if "FETCH_SIZE" in heads5 and "WRITE_SIZE" in heads5:
            t5["heads_hbm_bytes_per_launch"] = (2.0 * heads5["FETCH_SIZE"] + heads5["WRITE_SIZE"]) * 1024.0
        t5["algorithmic_bytes_per_forward"] = ("activations [169][256 hi | 256 lo] f16 = 173 KB per position, read and written "
                                                "by each of 38 layers + read as residual by 19: ~16.4 MB per position")
        json.dump(t5, open(os.path.join(summ, "config5_pmc_traffic.json"), "w"), indent=1)
    if per_fwd:
        d5 = {"kernel": "k_stem_wide_f16x3 + 38 x k_conv_wide_f16x3_s16, per leaf-batch forward (mean over the forwards of the leg's own counter runs)",
              "counters": per_fwd, "forwards": n_fwd, "positions_per_forward": pos}

        def fwd_ms(name):
            rows = [r for r in rows_of(name, "kernel_trace.csv") if is_wide_any(r["Kernel_Name"])]
            fw = sum(1 for r in rows if "k_stem_wide" in r["Kernel_Name"])
            return sum(dur_ms(r) for r in rows) / max(1, fw) if rows else None
        d5["forward_ms_kernel_trace_two_streams (sum of kernel durations in the driver's command; the streams overlap, so >= wall)"] = fwd_ms("stats")
        pipe(d5, per_fwd, fwd_ms("fetch_c5"), fwd_ms("sq1_c5"))
        json.dump(d5, open(os.path.join(summ, "config5_pmc_counters.json"), "w"), indent=1)

for f in ("resnet_pmc_counters.json", "resnet_pmc_traffic.json", "tree_pmc_traffic.json", "config5_pmc_traffic.json",
          "config5_pmc_counters.json", "kernel_durations.json"):
    p = os.path.join(summ, f)
    if os.path.exists(p):
        print(f, open(p).read())
