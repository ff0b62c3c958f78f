"""bench.py's host-side arithmetic (no GPU): the SURVEY 8(d) flop and byte models, the derived throughput
fields, and that the committed PMC traffic files are keyed to the command the driver runs -- so that
`roofline.traffic` is attached (not null) for `python bench.py --gpus 1 --steps 20 --warmup 5`."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_flop_model_matches_survey_8d():
    assert bench.flop_per_position(11, 6, 64) == 107851784 == bench.FLOP_PER_POSITION_6x64_11
    # 13x13, 19x256: stem 3 115 008 + 38 convs x 199 360 512 + heads
    assert bench.flop_per_position(13, 19, 256) == 3115008 + 38 * 199360512 + (173056 + 346112 + 43264 + 128 + 228488)


def test_byte_model_matches_survey_8d():
    # one simulation through D = 2 interior nodes with 100 and 99 children, a leaf with 98 legal moves
    st = dict(sum_depth=2, sum_k_interior=199, sum_k_leaf=98, selects=1, evals=1)
    select = (8 + 12 * 100) + (8 + 12 * 99)
    vloss = 32 * 2
    backup = 16 * (2 + 1)
    expand = 4 * 98 + 24 * 98 + 8
    assert bench.model_bytes(st) == select + vloss + backup + expand


def test_throughput_fields_are_consistent():
    f = bench.throughput_fields([4100.0, 2.0, 10.0, 4000.0, 150.0, 180.0, 0.0], 2.0, 5)
    assert f["value"] == 2050.0 and f["games_per_sec"] == 1.0 and f["plies_per_sec"] == 5.0
    assert f["mean_game_length"] == 90.0 and abs(f["games_per_sec_steady"] - 5.0 / 90.0) < 1e-12
    assert f["ms_per_step"] == 400.0 and f["games_finished"] == 2.0
    g = bench.throughput_fields([4100.0, 0.0, 10.0, 4000.0, 0.0, 0.0, 0.0], 2.0, 5)
    assert g["mean_game_length"] is None and g["games_per_sec_steady"] is None and g["games_per_sec"] == 0.0


def test_committed_pmc_traffic_is_keyed_to_the_drivers_command():
    # the defaults main() derives for `--steps 20 --warmup 5` on an 11x11 board
    a = argparse.Namespace(games=4096, board=11, sims=400, batch=10, blocks=6, chans=64, noise_scale=0.25,
                           desync=int(round(0.76 * 121)), settle=2 * 121, c5_games=512, nodes_per_game=0)
    a5 = bench.config5_args(a)
    assert (a5.board, a5.blocks, a5.chans, a5.sims, a5.games, a5.desync, a5.settle) == (13, 19, 256, 800, 512, 128, 338)
    key_r = [a.games, a.board, a.sims, a.batch, a.blocks, a.chans, 20, 5, a.noise_scale, a.desync, a.settle]
    key_t = [a.games, a.board, a.sims, a.batch, 130, 20, a.noise_scale, a.desync, a.settle]
    key_5 = [a5.games, a5.board, a5.batch, a5.blocks, a5.chans, "per forward"]
    for leg, key in (("resnet", key_r), ("tree", key_t), ("config5", key_5)):
        name = bench.PMC_FILES[leg]
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):       # recorded on the GPU box once per round (tools/prof.sh)
            continue
        t = json.load(open(path))
        assert t["bench_key"] == key, name
        assert t.get("src_sha"), "counter files are keyed to the kernel sources they were recorded on"
        kernels = "tree=...; net=...; src=%s" % t["src_sha"]
        traffic, src = bench.pmc_traffic(name, key, kernels)
        assert traffic and traffic == t["hbm_bytes_per_launch"] and name in src
        assert abs(t["hbm_bytes_per_launch"] - (2 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024) < 1.0
        assert bench.pmc_traffic(name, key[:-1] + [0], kernels)[0] is None
        # another library (other kernel sources) does not get these counters attached
        other, why = bench.pmc_traffic(name, key, "tree=...; src=0000000000000000")
        assert other is None and "not attached" in why


def test_src_sha_is_read_from_the_kernel_description():
    assert bench.src_sha("tree=k_play<2>; net=none; switches: A=0; src=207b9a732d64cae7") == "207b9a732d64cae7"
    assert bench.src_sha("tree=x") is None and bench.src_sha(None) is None


def test_cpu_baseline_carries_the_reference_shim_context():
    # BASELINE.md section 2, the rows bench.py quotes beside the port's figures
    r = bench.REFERENCE_SHIM["resnet"]
    assert r["sims_per_s"] == 3011.0 and r["cores"] == 8 and abs(r["per_core"] - 3011.0 / 8) < 0.1
    assert bench.REFERENCE_SHIM["tree"]["sims_per_s"] == 870.0


def _is_scalar(v):
    return v is None or isinstance(v, (bool, int, float, str))


def test_every_leg_s_fraction_can_be_recomputed_from_the_scalars_the_driver_keeps():
    """BENCH_r05.parsed showed what the driver keeps: the contract's top-level keys, and of `roofline` / `cpu_baseline` /
    `config` the SCALAR entries only (a nested `legs` dict was dropped).  So every non-scalar is
    stripped BEFORE anything is recomputed: the flat tree_* / c5_* / train_* / api_* / box_* keys must do alone."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r5_selfplay_bench.json")))
    line["roofline"].pop("legs", None)           # the r5 line carried the nested form
    line["cpu_baseline"].pop("legs", None)
    bench.flatten_legs(line)
    kept = bench.strip_to_driver_record(line)
    assert set(kept) <= {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
    for name in ("config", "roofline", "cpu_baseline"):
        assert all(_is_scalar(v) for v in kept[name].values()), name
    r, c = kept["roofline"], kept["cpu_baseline"]
    # headline (as before)
    assert abs(r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12 / r["peak"] - r["frac"]) < 1e-9
    assert kept["unit"] == "sims/s" and r["unit"] == "TFLOP/s"
    # configs[1]: model bytes and counter bytes over the launch time, against the HBM peak
    t = line["tree"]
    assert abs(r["tree_bytes_per_launch"] / (r["tree_avg_launch_ms"] * 1e-3) / 1e9 / r["tree_peak_gbs"] - t["roofline"]["frac"]) < 1e-9
    assert abs(r["tree_traffic"] / (r["tree_avg_launch_ms"] * 1e-3) / 1e9 / r["tree_peak_gbs"] - t["roofline"]["achieved_hbm_frac"]) < 1e-9
    assert r["tree_frac"] == t["roofline"]["frac"] and r["tree_sims_per_sec"] == t["value"] and r["tree_achieved_unit"] == "GB/s"
    assert r["tree_ms_per_move"] == t["roofline"]["ms_per_move"] and r["tree_steps"] == t["steps"]
    # configs[4]'s shape on one GPU
    c5 = line["config5"]
    assert abs(r["c5_flop_per_launch"] / (r["c5_avg_launch_ms"] * 1e-3) / 1e12 / r["c5_peak_tflops"] - c5["roofline"]["frac"]) < 1e-9
    assert r["c5_frac"] == c5["roofline"]["frac"] and r["c5_sims_per_sec"] == c5["value"] and r["c5_traffic"] == c5["roofline"]["traffic"]
    assert r["c5_positions_per_launch"] > 0 and r["c5_board"] == 13 and r["c5_games"] == 512 and r["c5_achieved_unit"] == "TFLOP/s"
    # both training steps
    ts = line["train_step"]
    assert abs(r["train_flop_per_step"] / (r["train_native_step_only_ms"] * 1e-3) / 1e12 / 2500.0 - r["train_native_frac"]) < 1e-9
    assert abs(r["train_wide_flop_per_step"] / (r["train_wide_native_step_only_ms"] * 1e-3) / 1e12 / 2500.0 - r["train_wide_frac"]) < 1e-9
    assert r["train_native_ms"] == ts["native"]["ms_per_step"] and r["train_hipgraph_ms"] == ts["hip_graph"]["ms_per_step"]
    assert r["train_eager_ms"] == ts["eager"]["ms_per_step"] and r["train_wide_hipgraph_ms"] == ts["wide"]["hip_graph"]["ms_per_step"]
    assert r["train_wide_native_ms"] == ts["wide"]["native"]["ms_per_step"]
    # product surface and the box yardstick
    assert r["api_rows_over_plies"] == line["api"]["rows_over_plies"] and r["api_rows_per_sec"] == line["api"]["rows_per_sec"]
    assert r["box_gemm_tflops"] == line["box"]["gemm_f16_8192_tflops"] and abs(r["box_relative"] - r["box_gemm_tflops"] / r["box_usual_tflops"]) < 1e-12
    assert r["games_per_sec"] == line["games_per_sec"]
    # the CPU side: every leg's baseline and the context that fixes its reading
    assert c["tree_value"] == t["cpu_baseline"]["value"] and c["tree_cores"] == 64 and c["tree_unit"] == "sims/s"
    assert c["c5_value"] == c5["cpu_baseline"]["value"] and c["c5_cores"] == 64
    assert c["reference_per_core"] == 376.4 and c["reference_scaled_value"] == 376.4 * c["cores"]
    assert c["same_container_port_per_core"] == 380.8 and c["same_container_reference_per_core"] == 376.4
    assert c["tree_reference_per_core"] == 870.0


def test_a_failed_leg_leaves_a_scalar_error():
    line = {"roofline": {"frac": 0.2}, "cpu_baseline": {"value": 1.0}, "config5": {"error": "RuntimeError('x')"},
            "train_step": {"error": "boom"}, "api": {"error": "nope"}}
    bench.flatten_legs(line)
    r = line["roofline"]
    assert r["c5_error"].startswith("RuntimeError") and r["train_error"] == "boom" and r["api_error"] == "nope"


def test_world_fields_of_an_n_gpu_line():
    """N > 1: distinct devices, per-rank spread, all-gather rate and efficiency as flat scalars."""
    per_rank = [{"rank": r, "device": "box|uuid-%d|%d" % (r % 4, r % 4), "sims_per_sec": 100.0 + r, "seconds": 1.0} for r in range(8)]
    x = {"bytes_gathered": 4096000, "allgather_seconds": 0.002}
    f = bench.world_fields(per_rank, 800.0, x, 110.0, "nccl")
    assert f["world_ranks"] == 8 and f["world_distinct_devices"] == 4 and f["rccl_ranks"] == 4      # two ranks per GPU: visible
    assert f["world_rank_sims_per_sec_min"] == 100.0 and f["world_rank_sims_per_sec_max"] == 107.0
    assert abs(f["weak_scaling_eff"] - 800.0 / (8 * 110.0)) < 1e-12 and abs(f["replay_allgather_gbs"] - 2.048) < 1e-9
    g = bench.world_fields(per_rank, 800.0, None, None, "gloo")
    assert g["rccl_ranks"] == 0 and "weak_scaling_eff" not in g and "replay_allgather_gbs" not in g
    assert all(v is None or isinstance(v, (bool, int, float, str)) for v in f.values())


def test_the_committed_round6_line_reproduces_every_leg_from_its_scalars():
    """profiles/r6_selfplay_bench.json is the line `python bench.py` printed on the MI355X box with the final library: cut
    down to what the driver's record keeps, it alone must reproduce the headline, tree, config5 and both training fractions,
    carry HBM traffic for all three kernel legs (PMC files keyed to the same kernel sources) and the train-loop leg."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r6_selfplay_bench.json")))
    kept = bench.strip_to_driver_record(line)
    r, c = kept["roofline"], kept["cpu_baseline"]
    assert abs(r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12 / r["peak"] - r["frac"]) < 1e-9 and 0.2 < r["frac"] < 1 / 3
    assert abs(r["tree_bytes_per_launch"] / (r["tree_avg_launch_ms"] * 1e-3) / 1e9 / r["tree_peak_gbs"] - r["tree_frac"]) < 1e-9
    assert abs(r["tree_traffic"] / (r["tree_avg_launch_ms"] * 1e-3) / 1e9 / r["tree_peak_gbs"] - r["tree_achieved_hbm_frac"]) < 1e-9
    assert abs(r["c5_flop_per_launch"] / (r["c5_avg_launch_ms"] * 1e-3) / 1e12 / r["c5_peak_tflops"] - r["c5_frac"]) < 1e-9
    assert r["c5_steps"] == 3 and r["c5_positions_per_launch"] > 4000
    assert abs(r["train_flop_per_step"] / (r["train_native_step_only_ms"] * 1e-3) / 1e12 / 2500.0 - r["train_native_frac"]) < 1e-9
    assert abs(r["train_wide_flop_per_step"] / (r["train_wide_native_step_only_ms"] * 1e-3) / 1e12 / 2500.0 - r["train_wide_frac"]) < 1e-9
    assert r["traffic"] > 0 and r["tree_traffic"] > 0 and r["c5_traffic"] > 0          # counters attached: same kernel sources
    assert r["traffic"] == r["traffic_tower"] + r["traffic_heads"]
    assert r["loop_inline_steps_per_sec"] > 0 and abs(r["loop_overlap_speedup"] - r["loop_overlapped_steps_per_sec"] / r["loop_inline_steps_per_sec"]) < 1e-9
    assert r["box_gemm_tflops"] > 0 and r["api_rows_over_plies"] > 0.9
    assert c["tree_value"] > 0 and c["c5_value"] > 0 and c["reference_scaled_value"] == c["reference_per_core"] * c["cores"]
    src = bench.src_sha(line["kernels"])
    for leg in ("resnet", "tree", "config5"):
        assert json.load(open(os.path.join(ROOT, "profiles", bench.PMC_FILES[leg])))["src_sha"] == src
