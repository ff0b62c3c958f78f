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


def test_tree_bench_line():
    d = _run("--steps", "3", "--warmup", "1", "--games", "256")
    _check_common(d, 3, 1)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    # 256 games x 3 moves x 410 select_leaf calls, exactly
    assert round(d["value"] * d["elapsed_s"]) == 256 * 3 * 410
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "sims/s" and c["sample"]


def test_resnet_bench_line():
    d = _run("--workload", "resnet", "--steps", "1", "--warmup", "1", "--games", "64", "--sims", "40",
             "--no-cpu-baseline")
    _check_common(d, 1, 1)
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["unit"] == "TFLOP/s"
    assert round(d["value"] * d["elapsed_s"]) == 64 * 1 * 50
