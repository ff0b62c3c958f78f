"""Perf floors of the two hot kernels: refactors around them -- the weight packers, the range guard in
the tower epilogues, the multi-rank plumbing -- must not silently slow the path the headline is measured on.  The
bounds sit 7-10 % above what the boxes of rounds 3-4 measured (tower + heads 7.95-8.03 ms per 40 960 positions, box to
box +-2 %, of which 0.7 % is this round's activation range tracking; 1.24-1.26 ms per k_play<2> move of 4096 games;
0.83-0.92 ms per hand-written training step)."""
import json
import os

import numpy as np
import pytest
import torch

from azalea_amd import engine as eng
from azalea_amd.network import HexNetwork

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("AZX_PERF_FLOOR") == "0",
                                 reason="AZX_PERF_FLOOR=0: wall-clock floors switched off (a shared or throttled box)")]
# Every floor is DEVICE time (HIP events on the stream the kernels run on), best of three measurements: host jitter
# and a neighbour's burst do not reach it, which is what lets the bounds sit 7-15 % above the measured values
# -- on a box of the usual speed; see box_factor below for the others.
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Boxes of this pool differ by more than the margins: the same library measured 7.93 ms per 40 960 positions on one
# lease and 9.07 on another an hour later -- and a plain hipBLASLt f16 GEMM (8192^3, torch.matmul) ran at 1 310 and
# 1 150 TFLOP/s on the same two (ratio 1.14 both ways: a power / clock difference of the box, not of the code).  So each
# floor is scaled by how far THIS box's GEMM falls short of the usual rate; on a box at or above it the floors are the
# plain numbers.  A box more than 35 % short is not measured at all.
TYPICAL_GEMM_TFLOPS = 1310.0
_box = {}


def box_factor():
    if "factor" not in _box:
        n, iters = 8192, 30
        a = torch.randn(n, n, device="cuda:0", dtype=torch.float16)
        b = torch.randn(n, n, device="cuda:0", dtype=torch.float16)
        for _ in range(5):
            a @ b
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                a @ b
            e1.record()
            torch.cuda.synchronize()
            best = max(best, iters * 2.0 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        del a, b
        torch.cuda.empty_cache()
        _box["gemm_tflops"] = best
        _box["factor"] = max(1.0, TYPICAL_GEMM_TFLOPS / best)
        _record("box", {"gemm_f16_8192_tflops": best, "typical": TYPICAL_GEMM_TFLOPS, "floor_scale": _box["factor"]})
    if _box["factor"] > 1.35:
        why = ("this box runs a library GEMM at %.0f TFLOP/s (usually %.0f): too throttled to hold a floor against -- the floors "
               "were NOT checked in this run" % (_box["gemm_tflops"], TYPICAL_GEMM_TFLOPS))
        _record("floors_skipped", why)                  # ... and say so where the numbers are read
        print("PERF FLOORS SKIPPED: " + why)
        pytest.skip(why)
    return _box["factor"]


def _record(name, value):
    path = os.path.join(ROOT, "gpurun_out", "perf_floor.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = value
    json.dump(data, open(path, "w"), indent=1)


def _pool(evaluator, **kw):
    E = eng.Engine(board_size=11, n_games=4096, simulations=400, search_batch_size=10, exploration_coef=0.5,
                   exploration_depth=15, noise_alpha=0.03, noise_scale=0.25, temperature=1.0, evaluator=evaluator, **kw)
    idx = np.arange(4096, dtype=np.int64)
    E.reset(moves=eng.random_prefixes(11, idx, 92, 0xBAD5EED5))       # the bench's de-synchronised start
    return E


def test_tower_and_heads_floor():
    f = box_factor()
    E = _pool(eng.EVAL_RESNET, num_blocks=6, base_chans=64)
    torch.manual_seed(0)
    net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).eval().to("cuda:0")
    E.set_weights({k: (v.data_ptr(), v.numel()) for k, v in net.state_dict().items() if v.dtype == torch.float32},
                  on_device=True)
    E.play_steps(1)
    ms_40960 = 1e9
    for _ in range(3):
        st = E.play_steps(2)
        ms = 1e3 * st["net_seconds"] / st["net_launches"]
        pos = st["evals"] / st["net_launches"]
        if ms * 40960.0 / pos < ms_40960:
            per_launch_ms, positions, ms_40960 = ms, pos, ms * 40960.0 / pos
    _record("tower_heads", {"ms_per_launch_pair": per_launch_ms, "positions_per_launch": positions,
                            "ms_per_40960_positions": ms_40960, "kernels": E.kernel_info()})
    E.close()
    assert positions > 30000
    assert ms_40960 <= 8.6 * f, (per_launch_ms, positions, f)


def test_tree_move_floor():
    f = box_factor()
    E = _pool(eng.EVAL_UNIFORM)
    E.play_steps(120)                                                   # settle: the pool spreads over all plies
    ms_per_move = 1e9
    for _ in range(3):
        st = E.play_steps(60)
        ms_per_move = min(ms_per_move, 1e3 * st["mcts_seconds"] / st["mcts_launches"])
    _record("tree", {"ms_per_move": ms_per_move, "moves": st["mcts_launches"], "kernel_launches": st["mcts_kernel_launches"],
                     "kernels": E.kernel_info()})
    E.close()
    assert st["mcts_kernel_launches"] < st["mcts_launches"]            # the persistent k_play path was taken
    assert ms_per_move <= 1.45 * f, (ms_per_move, f)


def test_native_training_step_floor():
    """The hand-written training step at the reference's shape (6x64, 11x11, batch 128): <= 0.55 ms per step with the
    inputs resident (measured 0.48-0.51 box to box; 0.83 before the split-f16 kernels; the stock kernels captured as a
    HIP graph take 2.1)."""
    f = box_factor()
    import time
    from azalea_amd.native_train import NativeTrainStep
    dev = "cuda:0"
    torch.manual_seed(0)
    net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).to(dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    B = 128
    step = NativeTrainStep(net, opt, B, dev)
    rng = np.random.RandomState(0)
    board = rng.randint(0, 3, (B, 11, 11)).astype(np.int32)
    board[rng.rand(B, 11, 11) < 0.4] = 0
    lm = np.zeros((B, 121), np.int32)
    mp = np.zeros((B, 121), np.float32)
    for i in range(B):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        lm[i, :len(e)] = e
        mp[i, :len(e)] = 1.0 / len(e)
    step.step(dict(board=torch.tensor(board, device=dev), legal_moves=torch.tensor(lm, device=dev),
                   moves_prob=torch.tensor(mp, device=dev), reward=torch.tensor(rng.choice([-1.0, 1.0], B).astype(np.float32), device=dev)))
    for _ in range(20):
        step._run()
    torch.cuda.synchronize()
    ms, host_ms = 1e9, 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()                       # the step's data chain runs on torch's current stream; its side stream joins it
        for _ in range(200):
            step._run()
        e1.record()
        torch.cuda.synchronize()
        host_ms = min(host_ms, 1e3 * (time.perf_counter() - t0) / 200)
        ms = min(ms, e0.elapsed_time(e1) / 200)
    _record("train_step", {"ms_per_step": ms, "steps_per_sec": 1e3 / ms, "host_clock_ms_per_step": host_ms})
    step.close()
    assert ms <= 0.55 * f, (ms, host_ms, f)


def test_wide_training_step_floor():
    """The hand-written step at BASELINE configs[4]'s network (19x256 on 13x13, batch 128): <= 10.9 ms per step with the
    inputs resident (measured 9.6-10.0 box to box on two streams with k_tw_wgrad2, 10.2-10.4 with k_tw_wgrad; the stock
    kernels captured as a HIP graph take 33)."""
    f = box_factor()
    from azalea_amd.native_train import NativeTrainStep
    dev = "cuda:0"
    torch.manual_seed(0)
    n, B = 13, 128
    net = HexNetwork(board_size=n, num_blocks=19, base_chans=256).to(dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    step = NativeTrainStep(net, opt, B, dev)
    rng = np.random.RandomState(0)
    cells = n * n
    board = rng.randint(0, 3, (B, n, n)).astype(np.int32)
    board[rng.rand(B, n, n) < 0.4] = 0
    lm = np.zeros((B, cells), np.int32)
    mp = np.zeros((B, cells), np.float32)
    for i in range(B):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        lm[i, :len(e)] = e
        mp[i, :len(e)] = 1.0 / len(e)
    step.step(dict(board=torch.tensor(board, device=dev), legal_moves=torch.tensor(lm, device=dev),
                   moves_prob=torch.tensor(mp, device=dev), reward=torch.tensor(rng.choice([-1.0, 1.0], B).astype(np.float32), device=dev)))
    for _ in range(3):
        step._run()
    torch.cuda.synchronize()
    ms = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step._run()
        e1.record()
        torch.cuda.synchronize()
        ms = min(ms, e0.elapsed_time(e1) / 10)
    _record("train_step_wide", {"ms_per_step": ms, "shape": "19x256 on 13x13, batch 128"})
    step.close()
    assert ms <= 10.9 * f, (ms, f)
