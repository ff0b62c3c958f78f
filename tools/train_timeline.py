#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace (tools/prof_train.sh with GRAPH=1): start offset, duration
and stream of every kernel of the last complete step -- who waits for whom.  usage: train_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "k_trn" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at k_trn_prep
starts = [i for i, r in enumerate(rows) if "k_trn_prep" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = t0
gap = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gap += (s - busy_end) / 1e3
    busy_end = max(busy_end, e)
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print("%8.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:50]))
print("step span %.1f us, idle gaps (no kernel running) %.1f us" % ((busy_end - t0) / 1e3, gap))
