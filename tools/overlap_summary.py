#!/usr/bin/env python3
"""Diagnostic: from a rocprofv3 kernel trace of the play-ahead loop, how much of the time the self-play kernels and the
training step's kernels were on the device TOGETHER.
    python3 tools/overlap_summary.py <kernel_trace.csv>
Families: self-play = k_tower_*, k_heads*, k_mcts*, k_choose, k_advance, k_play*, k_rows_pack, pack kernels; training =
k_trn_*, k_tw_*, k_replay_collate, k_records_put.  Prints per family: launches, summed device time, the share of that
time covered by at least one kernel of the OTHER family, and a sample of the timeline (one tower launch with every
training kernel that started inside it)."""
import csv
import sys


def family(name):
    if any(k in name for k in ("k_trn_", "k_tw_", "k_replay_collate", "k_records_put", "k_conv_wide_train")):
        return "train"
    if any(k in name for k in ("k_tower", "k_heads", "k_mcts", "k_choose", "k_advance", "k_play", "k_rows_pack", "k_pack",
                               "k_net_", "k_conv_wide", "k_stem", "k_reset", "k_gather")):
        return "selfplay"
    return None


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def covered(iv, cover):
    """Total length of the intervals `iv` that lies inside the (disjoint, sorted) intervals `cover`."""
    tot, j = 0, 0
    for a, b in sorted(iv):
        while j < len(cover) and cover[j][1] <= a:
            j += 1
        k = j
        while k < len(cover) and cover[k][0] < b:
            tot += max(0, min(b, cover[k][1]) - max(a, cover[k][0]))
            k += 1
    return tot


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    fam = {"train": [], "selfplay": []}
    names = {"train": {}, "selfplay": {}}
    for r in rows:
        f = family(r["Kernel_Name"])
        if f is None:
            continue
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        fam[f].append((a, b))
        short = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        d = names[f].setdefault(short, [0, 0])
        d[0] += 1
        d[1] += b - a
    # the window in which both sides ran (from the first training kernel after self-play started to the last)
    t0 = max(min(a for a, _ in fam["train"]), min(a for a, _ in fam["selfplay"]))
    t1 = min(max(b for _, b in fam["train"]), max(b for _, b in fam["selfplay"]))
    print("window with both sides active: %.1f ms" % ((t1 - t0) / 1e6))
    cov = {f: union([(max(a, t0), min(b, t1)) for a, b in fam[f] if b > t0 and a < t1]) for f in fam}
    for f, other in (("train", "selfplay"), ("selfplay", "train")):
        iv = [(max(a, t0), min(b, t1)) for a, b in fam[f] if b > t0 and a < t1]
        tot = sum(b - a for a, b in iv)
        both = covered(iv, cov[other])
        print("%-9s %6d launches, %8.1f ms of kernel time in the window, %5.1f %% of it with a %s kernel running too"
              % (f, len(iv), tot / 1e6, 100.0 * both / max(1, tot), other))
    busy = {f: sum(b - a for a, b in cov[f]) for f in cov}
    any_busy = sum(b - a for a, b in union([tuple(x) for x in cov["train"]] + [tuple(x) for x in cov["selfplay"]]))
    print("device busy (either side) %.1f %% of the window; self-play %.1f %%, training %.1f %%, both at once %.1f %%"
          % (100.0 * any_busy / (t1 - t0), 100.0 * busy["selfplay"] / (t1 - t0), 100.0 * busy["train"] / (t1 - t0),
             100.0 * (busy["selfplay"] + busy["train"] - any_busy) / (t1 - t0)))
    for f in ("selfplay", "train"):
        top = sorted(names[f].items(), key=lambda kv: -kv[1][1])[:6]
        print("%s kernels: %s" % (f, ", ".join("%s x%d avg %.1f us" % (k, v[0], v[1] / v[0] / 1e3) for k, v in top)))
    # sample: one tower launch from the middle of the window and the training kernels that started inside it
    towers = sorted((a, b) for r in rows if "k_tower" in r["Kernel_Name"]
                    for a, b in [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]))] if a > t0 and b < t1)
    if towers:
        a, b = towers[len(towers) // 2]
        inside = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""))
                        for r in rows if family(r["Kernel_Name"]) == "train" and a <= int(r["Start_Timestamp"]) < b)
        print("sample: one k_tower launch of %.2f ms; %d training kernels started inside it (first 12, us from its start):"
              % ((b - a) / 1e6, len(inside)))
        for s, e, n in inside[:12]:
            print("    +%8.1f .. +%8.1f  %s" % ((s - a) / 1e3, (e - a) / 1e3, n))


if __name__ == "__main__":
    main()
