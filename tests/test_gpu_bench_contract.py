"""bench.py's output contract (one JSON line, the keys the driver reads), on a small run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT,
                                  stderr=subprocess.DEVNULL, timeout=600).decode()
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _check_common(d, steps, warmup):
    assert d["metric"] == "mcts_sims_per_sec" and d["unit"] == "sims/s" and d["value"] > 0
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["ms_per_step"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and isinstance(d["dtype"], str)
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["achieved"] > 0 and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r


def test_default_line_is_the_selfplay_headline_with_the_tree_numbers_nested():
    """The default workload is the north-star headline (configs[2]: resnet-driven self-play) from a
    de-synchronised pool, with the configs[1] tree-only run nested as "tree"; small sizes here."""
    d = _run("--steps", "2", "--warmup", "1", "--games", "64", "--sims", "40", "--tree-steps", "6",
             "--tree-warmup", "2", "--board", "7", "--blocks", "1")
    _check_common(d, 2, 1)
    assert d["config"]["workload"].startswith("resnet self-play") and "de-synchronised" in d["config"]["start"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["unit"] == "TFLOP/s"
    assert round(d["value"] * d["elapsed_s"]) == 64 * 2 * 50
    assert d["plies"] == 64 * 2 and d["games_finished"] >= 0 and "games_per_sec_steady" in d
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "sims/s" and "mid-game" in c["sample"]
    x = d["replay_allgather"]
    assert x["ranks"] == 1 and x["record_bytes"] == 272 and sum(x["rows_per_rank"]) >= 1
    t = d["tree"]
    assert t["steps"] == 6 and t["warmup"] == 2 and t["roofline"]["bound"] == "hbm" and t["roofline"]["peak"] == 8000.0
    assert round(t["value"] * t["elapsed_s"]) == 64 * 6 * 50
    assert t["games_finished"] > 0 and t["mean_game_length"] > 0          # steady state: games do finish
    assert abs(t["games_per_sec_steady"] * t["mean_game_length"] - t["plies_per_sec"]) < 1e-6 * t["plies_per_sec"]
    assert t["cpu_baseline"]["kind"] == "port"


def test_tree_bench_line():
    d = _run("--workload", "tree", "--steps", "3", "--warmup", "1", "--games", "256")
    _check_common(d, 3, 1)
    assert d["config"]["workload"].startswith("BASELINE configs[1]")
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    # 256 games x 3 moves x 410 select_leaf calls, exactly
    assert round(d["value"] * d["elapsed_s"]) == 256 * 3 * 410
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "sims/s" and c["sample"]


def test_resnet_bench_line():
    d = _run("--workload", "resnet", "--steps", "1", "--warmup", "1", "--games", "64", "--sims", "40",
             "--no-cpu-baseline", "--desync", "0")
    _check_common(d, 1, 1)
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and "lock-step" in d["config"]["start"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["unit"] == "TFLOP/s"
    assert round(d["value"] * d["elapsed_s"]) == 64 * 1 * 50
