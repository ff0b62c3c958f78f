"""bench.py's output contract (one JSON line, the keys the driver reads), on a small run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


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
    # the box's own speed by a library GEMM, in the line and inside the dict the driver keeps whole
    assert 100.0 < d["box"]["gemm_f16_8192_tflops"] < 2500.0
    assert "legs" not in r and r["box_gemm_tflops"] == d["box"]["gemm_f16_8192_tflops"]
    # the driver keeps only the scalars of roofline / cpu_baseline: every leg of the line is repeated there, flat
    kept = bench.strip_to_driver_record(d)
    for name in ("tree", "config5"):
        pre = {"tree": "tree_", "config5": "c5_"}[name]
        if name in d and "roofline" in d[name]:
            assert kept["roofline"][pre + "frac"] == d[name]["roofline"]["frac"]
            assert kept["roofline"][pre + "sims_per_sec"] == d[name]["value"]
            if "cpu_baseline" in d[name] and "cpu_baseline" in kept:
                assert kept["cpu_baseline"][pre + "value"] == d[name]["cpu_baseline"]["value"]


def test_default_line_is_the_selfplay_headline_with_the_tree_numbers_nested():
    """The default workload is the north-star headline (configs[2]: resnet-driven self-play) from a
    de-synchronised pool, with the configs[1] tree-only run nested as "tree"; small sizes here."""
    d = _run("--steps", "2", "--warmup", "1", "--games", "64", "--sims", "40", "--tree-steps", "6",
             "--tree-warmup", "2", "--board", "7", "--blocks", "1", "--api-moves", "70")
    _check_common(d, 2, 1)
    assert "config5" not in d and "train_step" not in d        # only nested under the 11x11 / 6x64 headline
    a = d["api"]                                               # the product surface, a game length after the transplant
    assert a["moves_since_transplant"] >= 70 and a["rows"] > 0 and d["rows_per_sec"] == a["rows_per_sec"] > 0
    assert 0.5 < a["rows_over_plies"] < 1.5 and 0 <= a["host_overhead_frac"] < 1
    assert {"search_root_visits", "search_root_children", "search_tree_nodes", "games"} <= set(a["metric_keys"])
    w = d["world"]
    assert (w["ranks"], w["backend"], w["games_per_rank"]) == (1, None, 64) and w["rows_per_rank"] == d["replay_allgather"]["rows_per_rank"]
    assert "k_mcts<2" in d["kernels"] and "k_heads" in d["kernels"] and "AZX_MCTS_GENERIC=0" in d["kernels"]
    assert d["config"]["workload"].startswith("resnet self-play") and "de-synchronised" in d["config"]["start"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["unit"] == "TFLOP/s"
    assert round(d["value"] * d["elapsed_s"]) == 64 * 2 * 50
    assert d["plies"] == 64 * 2 and d["games_finished"] >= 0 and "games_per_sec_steady" in d
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "sims/s" and "mid-game" in c["sample"]
    assert abs(c["per_core"] * c["cores"] - c["value"]) < 1e-6 * c["value"] and "reference_shim" not in c   # not the 11x11 / 6x64 net
    x = d["replay_allgather"]
    assert x["ranks"] == 1 and x["record_bytes"] == 272 and sum(x["rows_per_rank"]) >= 1
    t = d["tree"]
    assert t["steps"] == 6 and t["warmup"] == 2 and t["roofline"]["bound"] == "hbm" and t["roofline"]["peak"] == 8000.0
    assert round(t["value"] * t["elapsed_s"]) == 64 * 6 * 50
    assert t["games_finished"] > 0 and t["mean_game_length"] > 0          # steady state: games do finish
    assert abs(t["games_per_sec_steady"] * t["mean_game_length"] - t["plies_per_sec"]) < 1e-6 * t["plies_per_sec"]
    assert t["cpu_baseline"]["kind"] == "port" and t["cpu_baseline"]["reference_shim"]["per_core"] == 870.0
    assert "k_play<2> (persistent)" in t["kernels"]


def test_default_line_nests_the_config5_shape_with_roofline_and_cpu_baseline():
    """Under the real headline shape (11x11, 6x64) the line also carries BASELINE configs[4]'s shape on one GPU as
    "config5" -- 13x13, 19x256, 810 select_leaf calls, one warm-up + three timed moves -- with its own roofline and
    cpu_baseline (few games here)."""
    d = _run("--steps", "1", "--warmup", "1", "--games", "32", "--tree-steps", "4", "--tree-warmup", "1",
             "--c5-games", "8", "--api-moves", "0", "--settle", "30", "--loop-steps", "0")
    _check_common(d, 1, 1)
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and "api" not in d and "train_loop" not in d
    c = d["cpu_baseline"]
    assert c["reference_shim"]["sims_per_s"] == 3011.0 and c["reference_shim"]["cores"] == 8
    assert abs(c["port_vs_reference_per_core"] - c["per_core"] / c["reference_shim"]["per_core"]) < 1e-9
    c5 = d["config5"]
    assert c5["config"]["workload"].startswith("BASELINE configs[4] shape") and c5["steps"] == 3 and c5["warmup"] == 1
    assert round(c5["value"] * c5["elapsed_s"]) == 3 * 8 * 810 and c5["plies"] == 3 * 8
    r = c5["roofline"]
    assert r["bound"] == "mfma" and "k_conv_wide_f16x3_s16 x 38" in r["kernel"] and r["peak"] == 2500.0
    assert abs(r["flop_per_launch"] / r["positions_per_launch"] - 7.58e9) < 0.01e9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    b = c5["cpu_baseline"]
    assert b["kind"] == "port" and b["value"] > 0 and "bounded sample" in b["sample"] and "19x256" in b["sample"]
    assert "k_conv_wide_f16x3_s16" in c5["kernels"]
    ts = d["train_step"]          # beside the path: SURVEY 8(f).4 -- eager (synced / unsynced), captured, hand-written
    assert ts["batch"] == 128 and all(ts[m]["steps_per_sec"] > 0 for m in ("eager", "eager_nosync", "hip_graph", "native_fp32", "native"))
    assert abs(ts["speedup_native_vs_hip_graph"] - ts["native"]["steps_per_sec"] / ts["hip_graph"]["steps_per_sec"]) < 1e-9
    assert abs(ts["speedup_hip_graph_vs_eager_nosync"] - ts["hip_graph"]["steps_per_sec"] / ts["eager_nosync"]["steps_per_sec"]) < 1e-9
    n = ts["native"]
    assert n["step_only_ms"] > 0 and abs(n["vs_fp32_mfma_peak"] - n["algorithmic_tflops"] / 157.3) < 1e-6
    assert abs(n["frac_of_f16_mfma_peak"] - n["algorithmic_tflops"] / 2500.0) < 1e-9
    assert ts["native"]["steps_per_sec"] > 2.5 * ts["hip_graph"]["steps_per_sec"]      # measured 3.7-3.9x
    assert ts["native"]["steps_per_sec"] > 1.1 * ts["native_fp32"]["steps_per_sec"]    # the split-f16 kernels: measured 1.3-1.5x
    assert "fp32" in ts["native_fp32"]["arithmetic"] and "split f16" in ts["native"]["arithmetic"]
    kr = bench.strip_to_driver_record(d)["roofline"]      # ... and the same from the scalars the driver keeps
    assert abs(kr["train_flop_per_step"] / (kr["train_native_step_only_ms"] * 1e-3) / 1e12 / 2500.0 - kr["train_native_frac"]) < 1e-9
    assert abs(kr["train_wide_flop_per_step"] / (kr["train_wide_native_step_only_ms"] * 1e-3) / 1e12 / 2500.0 - kr["train_wide_frac"]) < 1e-9
    assert kr["train_hipgraph_ms"] == ts["hip_graph"]["ms_per_step"] and kr["train_wide_hipgraph_ms"] > kr["train_wide_native_ms"]
    assert abs(kr["c5_flop_per_launch"] / (kr["c5_avg_launch_ms"] * 1e-3) / 1e12 / kr["c5_peak_tflops"] - kr["c5_frac"]) < 1e-9
    assert abs(kr["tree_bytes_per_launch"] / (kr["tree_avg_launch_ms"] * 1e-3) / 1e9 / kr["tree_peak_gbs"] - kr["tree_frac"]) < 1e-9


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
