#!/usr/bin/env python3
"""Diagnostic: the whole-game distribution comparison at the headline's hyper-parameters (tests/test_gpu_game_distribution.py:
11x11, alpha 0.03, eps 0.25, depth 15) for several further INDEPENDENT engine seeds -- uniform-hash (512 games, 100 -> 110
selects, the oracle played live) and the 6x64 device network (256 games, 60 -> 70 selects, against fixture G13).
    python3 tools/dist_headline_seeds.py [n_seeds] > profiles/r6_distribution_headline_seeds.json"""
import json
import os
import pathlib
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import game_stats as gs                      # noqa: E402
import test_gpu_game_distribution as T       # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tmp = pathlib.Path(tempfile.mkdtemp())
cfg, plies = T.CFG_HEADLINE, T.HEADLINE_PLIES
out = {"what": "engine samples on further seed bases a million apart against the pooled oracle samples; two-sample tests of "
               "tests/game_stats.py; under the null ~5 % of the tests fall below 0.05 and the smallest p of n tests is ~1/n",
       "uniform_hash": [], "device_network": []}
a = T.oracle_sample(tmp, 11, 100, 512, 0, (), cfg)
b = T.oracle_sample(tmp, 11, 100, 512, 100000, (), cfg)
ref = {k: np.concatenate([a[k], b[k]]) for k in a}
for i in range(n_seeds):
    e = T.engine_sample(11, 100, 512, seed=31000000 + 1000000 * i, CFG=cfg)
    res = gs.compare(e, {k: ref[k] for k in e}, cfg["depth"], plies, 150)
    out["uniform_hash"].append({"seed": 31000000 + 1000000 * i, "tests": len(res), "below_0.05": int(sum(v < 0.05 for v in res.values())),
                                "min_p": min(res.values()), "worst": gs.worst(res)[0], "mean_length": float(e["length"].mean())})
z = np.load(os.path.join(ROOT, "tests", "golden", "g3_forward_11_6x64.npz"))
state = {k[2:]: z[k] for k in z.files if k.startswith("w:") and z[k].dtype.kind == "f"}
(oa, ob), _ = T.golden_oracle_games("g13_oracle_net_games_11h.npz")
for i in range(n_seeds):
    e = T.engine_sample(11, 60, 256, seed=41000000 + 1000000 * i, net=(6, 64, state), CFG=cfg)
    refn = {k: np.concatenate([oa[k], ob[k]]) for k in e}
    res = gs.compare(e, refn, cfg["depth"], plies, 80)
    out["device_network"].append({"seed": 41000000 + 1000000 * i, "tests": len(res), "below_0.05": int(sum(v < 0.05 for v in res.values())),
                                  "min_p": min(res.values()), "worst": gs.worst(res)[0], "mean_length": float(e["length"].mean())})
for k in ("uniform_hash", "device_network"):
    rows = out[k]
    out[k + "_summary"] = {"samples": len(rows), "tests": int(sum(r["tests"] for r in rows)), "below_0.05": int(sum(r["below_0.05"] for r in rows)),
                           "expected_below_0.05": 0.05 * sum(r["tests"] for r in rows), "smallest_p": min(r["min_p"] for r in rows)}
print(json.dumps(out, indent=1))
