"""Golden G12: the reference's Elo ranking (azalea/ranking.py:46-58) on a few tournaments' tallies.
Run in the build container only (imports /root/reference):  python tests/golden/make_ranking.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools", "refshim"))
from azalea import ranking  # noqa: E402

CASES = [
    (2, {(0, 1): (63, 0, 137)}),
    (3, {(0, 1): (12, 0, 8), (0, 2): (5, 0, 15), (1, 2): (9, 0, 11)}),
    (4, {(0, 1): (30, 0, 70), (0, 2): (20, 0, 80), (1, 2): (45, 0, 55), (0, 3): (10, 0, 90), (1, 3): (35, 0, 65),
         (2, 3): (40, 0, 60)}),
    (6, {(5, 3): (74, 0, 26), (4, 3): (66, 0, 34), (4, 2): (74, 0, 26), (2, 1): (60, 0, 40), (1, 0): (55, 0, 45),
         (5, 0): (97, 0, 3), (3, 0): (80, 0, 20)}),
    (3, {(1, 0): (7, 0, 3), (2, 1): (6, 0, 4), (0, 2): (2, 0, 8), (0, 1): (4, 0, 6)}),
]

out = []
for n, oc in CASES:
    scores = ranking.compute_ranking(n, oc)
    out.append({"players": n, "outcomes": [[list(k), list(v)] for k, v in oc.items()], "elo": [float(x) for x in scores]})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g12_ranking.json")
json.dump(out, open(path, "w"), indent=1)
print(path, [np.round(c["elo"], 1).tolist() for c in out])
