#!/usr/bin/env python3
"""bench.py -- self-play throughput of the HIP engine (games/s + MCTS sims/s, 11x11 Hex @ 400 sims).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload selfplay|resnet|tree]

A "step" is one engine move: every one of the `--games` concurrent games runs a full
400-simulation search ((400//10+1)*10 = 410 select_leaf calls, mcts.py:268), draws its move on
the device and advances; finished games are harvested whole and their slots restart in place.
Inputs are synthetic (random-init weights, device RNG) and already resident in HBM when the timed
region starts.

Default workload = the north-star headline, BASELINE.json configs[2]: 4096 concurrent 11x11 games,
400 sims/move, 6x64 resnet forward on the MFMA cores (MFMA-bound; `roofline` prices the tower +
heads launch pair).  The tree-only sub-benchmark, configs[1] (uniform priors, no net: HBM model of
SURVEY 8(d)), is measured the same way right after it and nested in the same line as "tree".
`--workload resnet` / `--workload tree` run one of the two alone (profiling).  Two more legs are nested in the
default line: "config5" -- BASELINE configs[4]'s shape on this one GPU (13x13, 19x256 resnet, 800 -> 810 sims,
`--c5-games` games, one warm-up and `--c5-steps` timed moves) with its own `roofline` (the wide tower's per-layer MFMA
kernel) and `cpu_baseline` -- and "api": the product surface the trainer calls (Player.read, parallel_player.py:
24-28) driven for `--api-moves` engine moves after the pool transplant, rows/s over the last 60 of them beside
the plies/s the engine played in the same window.  A last nested leg stands beside the path, not on it:
"train_step" -- the reference's training step at its own batch size on the HBM replay ring, eager and captured as a
HIP graph (`--no-train-step` skips it) -- and "train_loop": the trainer's loop on this GPU with the refills
played inline against self-play running beside the steps (azalea_amd/play_ahead.py; `--loop-steps 0` skips it).

Steady state: a pool that restarts finished games in place is, after its first game, spread over
all plies.  Starting every slot from the empty board would time the opening only (no game can end
in the first 2N-1 moves), so before the warm-up the pool is put into its steady state, untimed:
every slot starts from a seeded random legal position of 0..`--desync` plies and the cheap
uniform-prior self-play (the configs[1] engine) then plays `--settle` moves per slot -- two to three
game lengths, finished games restarting in place -- after which the resnet engine's slots are reset
to those self-played positions (`--desync 0` keeps the lock-step start from the empty board).
`games_per_sec` = games finished inside the timed region / its duration; `games_per_sec_steady` =
plies/s over the mean length of those games -- the two agree when the pool is in steady state.

N>1 is launched by the driver with torch.distributed.run, one rank per GPU: games shard across
ranks by global game index (rank r of W plays games r, r+W, ...: a fixed seed plays the same games
whatever W is) with no data-path collective (weak scaling); the barrier + max-over-ranks timing
use RCCL.  After the timed region the ranks exchange one replay refill the way a shared
DeviceReplayBuffer.consume does (records packed on the device, all-gather, ring put) and report
its duration as `replay_allgather`.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# committed rocprofv3 --pmc summaries of this round (tools/prof.sh); attached only when their bench_key matches AND they
# were recorded on the kernel sources this library was built from (src_sha, azx_kernel_info's `src=`)
PMC_FILES = {"resnet": "r6_resnet_pmc_traffic.json", "tree": "r6_tree_pmc_traffic.json",
             "config5": "r6_config5_pmc_traffic.json"}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TF = 157.3   # dense fp32 MFMA peak
F16_MFMA_PEAK_TF = 2500.0  # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline is 2:1 sparse)
FLOP_PER_POSITION_6x64_11 = 107851784


def flop_per_position(n, blocks, chans):
    """algorithmic flops of one evaluated position (2 per MAC; BN, ReLU, embedding,
    softmax excluded): stem conv 3x3 4->C, 2*blocks convs 3x3 C->C, value/policy 1x1 convs, the
    three FC layers.  (11, 6, 64) -> 107 851 784; (13, 19, 256) -> 7.58e9."""
    cells = n * n
    stem = cells * chans * 36 * 2
    tower = 2 * blocks * cells * chans * chans * 9 * 2
    heads = cells * chans * 2 * 2 + cells * chans * 4 * 2 + 2 * cells * 64 * 2 + 64 * 2 + 4 * cells * cells * 2
    return stem + tower + heads


def model_bytes(st):
    """SURVEY 8(d) algorithmic HBM bytes of the tree kernels (reference six-array data model):
    select: sum over scored interior nodes of (8 + 12 k_i); virtual loss apply+undo 32 D;
    backup 16 (D+1); expand 4 k_L read + 24 k_L + 8 written."""
    D, ki, kl = st["sum_depth"], st["sum_k_interior"], st["sum_k_leaf"]
    sel, ev = st["selects"], st["evals"]
    return 8 * D + 12 * ki + 32 * D + 16 * (D + sel) + 28 * kl + 8 * ev


# the reference itself (pure-Python rules through the numba shim, torch CPU conv), timed in the
# survey container on 8 cores -- it cannot travel to the GPU box, so it is carried here as context
REFERENCE_SHIM = {
    "resnet": {"sims_per_s": 3011.0, "cores": 8, "per_core": 376.4, "games_per_s": 0.068,
               "single_thread_sims_per_s": 437.0,
               "what": "reference parallel_player.Player + ProcessPool(8), 11x11, 400 sims, 6x64 (BASELINE.md section 2; "
                       "measured in the survey container, not on this box)"},
    "tree": {"sims_per_s": 870.0, "cores": 1, "per_core": 870.0,
             "what": "reference MCTS only, stub net (uniform priors), 400 sims from the empty board, 1 process, "
                     "pure-Python rules (BASELINE.md section 2; survey container)"},
}


def cpu_baseline(args, workload, net_state=None, sims=None, max_plies=None, games=None):
    """The CPU oracle (oracle/, a C restatement pinned against the reference's golden vectors)
    timed on this box's host cores on a bounded sample of the same workload.  `per_core` puts it beside the
    reference's own CPU path (`reference_shim`, BASELINE.md section 2: torch's conv2d, measured elsewhere)."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    sims = args.sims if sims is None else sims
    if workload == "tree":
        games, max_plies, net, start = 8 * threads, 300, None, 0      # ~5 s on 64 threads
        sample = "%d whole games, uniform priors, %d sims/move, one game per thread" % (games, sims)
    else:
        games = threads if games is None else games
        max_plies, start = (4 if max_plies is None else max_plies), args.desync   # ~13 s on 64 threads
        net = orc.Net(args.board, args.blocks, args.chans, net_state)
        sample = ("%d games x %d plies each from seeded random mid-game positions (0..%d stones, like the "
                  "GPU pool), %dx%d resnet fp32 on the host (blocked direct conv), %d sims/move, one game per thread"
                  % (games, max_plies, start, args.blocks, args.chans, sims))
        if sims != args.sims:
            sample += (" -- a bounded sample: the GPU leg searches %d sims/move, at the oracle's rate one such ply "
                       "would take minutes per thread; sims/s is set by the network forward either way" % args.sims)
    out = orc.bench_selfplay(args.board, sims, args.batch, games, threads, net=net,
                             max_plies=max_plies, seed=args.seed, start_max=start)
    sims_per_s = out["selects"] / out["seconds"]
    base = {"value": sims_per_s, "unit": "sims/s", "cores": threads, "per_core": sims_per_s / threads,
            "kind": "port", "sample": sample, "seconds": out["seconds"], "plies": out["plies"]}
    if workload == "tree":
        base["games_per_s"] = out["games"] / out["seconds"]
    else:
        base["plies_per_s"] = out["plies"] / out["seconds"]
    ref = REFERENCE_SHIM.get("tree" if workload == "tree" else "resnet")
    if ref and (workload == "tree" or (args.board, args.blocks, args.chans) == (11, 6, 64)):
        base["reference_shim"] = dict(ref)
        base["port_vs_reference_per_core"] = base["per_core"] / ref["per_core"]
        # what the reference itself would do on THIS box if it scaled linearly to these threads at its per-core rate of
        # the survey container: an upper estimate, quoted so that no GPU/CPU ratio is read off the port alone
        base["reference_scaled"] = {"value": ref["per_core"] * threads, "unit": "sims/s", "cores": threads,
                                    "how": "reference_shim.per_core (survey container) x this box's threads: not measured here"}
        if workload != "tree":
            base["same_container"] = dict(SAME_CONTAINER)
            base["note"] = ("per_core on this box (%d threads) is below the reference's per-core figure of the survey container, "
                            "but on that container the port does %.0f sims/s per core against the reference's %.0f: the two "
                            "are equal per core on equal hardware" % (threads, SAME_CONTAINER["port_per_core_8"],
                                                                       SAME_CONTAINER["reference_per_core_8"]))
    return base


PMC_EXTRA = {}     # file name -> further per-launch figures of the counter file last attached (k_heads bytes)


def src_sha(kernels):
    """`src=<digest>` of azx_kernel_info: the kernel sources the loaded library was built from."""
    for part in (kernels or "").split(";"):
        part = part.strip()
        if part.startswith("src="):
            return part[4:]
    return None


def pmc_traffic(name, key, kernels=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
    command (tools/prof.sh: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 read correction of
    MI355X_MICROARCH.md).  Only attached when the file was recorded for exactly this configuration and on exactly
    these kernel sources; otherwise (None, why)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no committed counter file (profiles/%s)" % name
    t = json.load(open(path))
    if t.get("bench_key") != key:
        return None, "profiles/%s was recorded for another configuration" % name
    sha = src_sha(kernels)
    if t.get("src_sha") != sha:
        return None, ("profiles/%s was recorded on kernel sources %s, this library is %s: not attached"
                      % (name, t.get("src_sha"), sha))
    PMC_EXTRA[name] = {k: t[k] for k in ("heads_hbm_bytes_per_launch",) if t.get(k) is not None}
    return t.get("hbm_bytes_per_launch"), "profiles/%s (PMC, same command, kernel sources %s)" % (name, sha)


def make_engine(workload, args, rank, world, local_rank):
    from azalea_amd import engine as eng
    evaluator = eng.EVAL_UNIFORM if workload == "tree" else eng.EVAL_RESNET
    return eng.Engine(board_size=args.board, n_games=args.games, simulations=args.sims,
                      search_batch_size=args.batch, exploration_coef=0.5, exploration_depth=15,
                      noise_alpha=0.03, noise_scale=args.noise_scale, temperature=1.0, evaluator=evaluator,
                      num_blocks=args.blocks, base_chans=args.chans, device=local_rank,
                      nodes_per_game=args.nodes_per_game, seed=args.seed,
                      game_index_stride=world, game_index_offset=rank)


def settle_pool(E, args, rank, world, local_rank):
    """Untimed: random start positions, then `--settle` moves of uniform-prior self-play per slot on the
    tree-only engine E, which leaves the pool in its steady state (slots spread over all plies the way a
    long-running pool is)."""
    import numpy as np
    from azalea_amd import engine as eng
    idx = np.arange(args.games, dtype=np.int64) * world + rank      # global game indices of the slots
    E.reset(moves=eng.random_prefixes(args.board, idx, args.desync, args.seed, device=local_rank))
    if args.settle:
        E.play_steps(args.settle)


def pool_positions(E):
    """The pool's current positions as move lists (any alternating order of a position's stones is a legal
    way to reach it: a position that is not won has no won sub-position)."""
    import numpy as np
    gm = E.get_games()
    assert not gm["result"].any()
    moves = []
    for b in gm["board"].reshape(len(gm["board"]), -1):
        xs, os_ = np.flatnonzero(b == 1) + 1, np.flatnonzero(b == 2) + 1
        assert len(xs) - len(os_) in (0, 1)
        seq = np.empty(len(xs) + len(os_), np.int64)
        seq[0::2], seq[1::2] = xs, os_
        moves.append(seq.tolist())
    return moves


def run_workload(workload, args, rank, world, local_rank, steps, warmup, sync, torch):
    """One engine, its pool put into steady state (untimed), warmed up and timed for `steps` moves.
    Returns (st, elapsed, extras)."""
    start = None
    if args.desync > 0 and workload != "tree":
        T = make_engine("tree", args, rank, world, local_rank)
        settle_pool(T, args, rank, world, local_rank)
        start = pool_positions(T)
        T.close()
    E = make_engine(workload, args, rank, world, local_rank)
    extras = {"net_state": None}
    if workload == "resnet":
        from azalea_amd.network import HexNetwork
        torch.manual_seed(0)
        net = HexNetwork(board_size=args.board, num_blocks=args.blocks, base_chans=args.chans)
        net.eval().to("cuda:%d" % local_rank)
        sd = net.state_dict()
        E.set_weights({k: (v.data_ptr(), v.numel()) for k, v in sd.items()
                       if v.dtype == torch.float32}, on_device=True)
        extras["net_state"] = {k: v.detach().cpu().numpy() for k, v in sd.items()}
    if start is not None:
        E.reset(moves=start)
    elif args.desync > 0:
        settle_pool(E, args, rank, world, local_rank)
    if warmup:
        E.play_steps(warmup)
    sync()
    t0 = time.perf_counter()
    st = E.play_steps(steps)          # blocks until the engine stream has drained
    sync()
    elapsed = time.perf_counter() - t0
    extras["engine"] = E
    extras["start"] = start
    extras["kernels"] = E.kernel_info()
    return st, elapsed, extras


def config5_args(args):
    """BASELINE configs[4]'s board, network and search on ONE GPU: 13x13, 19x256, 800 sims/move (810 select_leaf
    calls); `--c5-games` concurrent games (the 8-GPU config shards 8x as many)."""
    a = argparse.Namespace(**vars(args))
    a.board, a.blocks, a.chans, a.sims, a.games = 13, 19, 256, 800, args.c5_games
    a.desync = int(round(0.76 * a.board * a.board))
    a.settle = 2 * a.board * a.board
    a.nodes_per_game = 0
    return a


TRAIN_FLOP_PER_BOARD_CONV = None


def train_step_flops(n, blocks, chans, batch):
    """Algorithmic flops of one training step (2 per MAC): forward = flop_per_position per board, backward = twice that
    (data + weight gradients of every layer); BatchNorm, ReLU, loss and the SGD update excluded."""
    return 3.0 * batch * flop_per_position(n, blocks, chans)


TRAIN_MODES = ("eager", "eager_nosync", "hip_graph", "native_fp32", "native")


def train_step_mode(args, local_rank, torch, mode):
    """One mode of the training-step leg in THIS process (the parent runs each mode in a process of its own: the modes
    leave state behind -- torch side streams, MIOpen workspaces, allocator pools -- that costs whichever runs next
    20-50 %)."""
    import numpy as np
    from torch import optim
    from azalea_amd import engine as eng
    from azalea_amd.device_replay import DeviceReplayBuffer
    from azalea_amd.native_train import NativeTrainStep
    from azalea_amd.network import HexNetwork
    from azalea_amd.policy_trainer import GraphedTrainStep, supervised_step
    dev = torch.device("cuda", local_rank)
    B, steps = 128, args.train_steps
    warm = max(3, steps // 10)
    wide = args.chans >= 128
    E = eng.Engine(board_size=args.board, n_games=1024, simulations=50, search_batch_size=10,
                   evaluator=eng.EVAL_UNIFORM, noise_scale=0.25, device=local_rank)
    buf = DeviceReplayBuffer(E, 20000, shared=False)
    E.replay_fill(20000)
    torch.manual_seed(0)
    order = np.resize(buf.epoch_indices(), (steps + warm) * B)
    net = HexNetwork(board_size=args.board, num_blocks=args.blocks, base_chans=args.chans).to(dev)
    opt = optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    gs = (GraphedTrainStep(net, opt, B, dev) if mode == "hip_graph" else
          NativeTrainStep(net, opt, B, dev) if mode in ("native", "native_fp32") else None)
    n_steps = steps if gs is not None else min(steps, 60)
    for i in range(n_steps + warm):
        if i == warm:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        idx = order[i * B:(i + 1) * B]
        if gs is not None:
            gs.step_from_ring(buf, idx)            # rows collated straight into the step's static inputs
        elif mode == "eager":
            supervised_step(net, buf.sample(idx), train=True, optimizer=opt, device=dev)
        else:                                      # the same eager kernels, the loss left on the device
            batch = buf.sample(idx)
            net.train()
            opt.zero_grad()
            o = net.forward(batch["board"], batch["legal_moves"])
            loss = (torch.nn.functional.mse_loss(o["value"], batch["reward"])
                    - (batch["moves_prob"] * o["moves_logprob"]).sum() / B)
            loss.backward()
            opt.step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    out = {"steps_per_sec": n_steps / dt, "ms_per_step": 1e3 * dt / n_steps, "positions_per_sec": n_steps * B / dt,
           "steps": n_steps, "ring_rows": len(buf)}
    if mode in ("native", "native_fp32"):
        # the step alone (inputs resident, no collate): HIP events on torch's stream around `steps` launches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            gs._run()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        fl = train_step_flops(args.board, args.blocks, args.chans, B)
        out.update({"step_only_ms": ms, "step_only_steps_per_sec": 1e3 / ms,
                    "algorithmic_tflops": fl / (ms * 1e-3) / 1e12,
                    "arithmetic": ("exact fp32 on v_mfma_f32_32x32x2_f32 (AZX_TRAIN_FWD / _BWD / _WGRAD=fp32)" if mode == "native_fp32" else
                                   "split f16 (hi, lo) x3 on v_mfma_f32_32x32x16_f16, fp32 accumulate, operands scaled per layer "
                                   "by powers of two; AZX_TRAIN_FWD / _BWD / _WGRAD=fp32 select the exact-fp32 MFMA kernels"),
                    "frac_of_f16_mfma_peak": fl / (ms * 1e-3) / 1e12 / 2500.0,
                    "issued_frac_of_f16_mfma_peak": 3.0 * fl / (ms * 1e-3) / 1e12 / 2500.0,
                    "vs_fp32_mfma_peak": fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF,
                    "bound": ("throughput: per layer three split-f16 MFMA convolutions (k_conv_wide_train forward and backward-data, "
                              "k_tw_wgrad filter gradient) and two elementwise passes (DESIGN 8.5)") if wide else
                             "launch chain: ~31 dependent kernels of 12-16 us on the data stream whose matrix work is ~1.5 us each (DESIGN 8.4)",
                    "flop_per_step": fl})
        gs.close()
    E.close()
    return out


def run_train_step(args, local_rank, torch):
    """SURVEY 8(f).4 beside the path: the reference's training step (policy_trainer.py:123-142) at its own batch of 128
    (config/hex11_train_config.yml), fed from the HBM replay ring, four ways: as the reference runs it (eager, a host
    sync per step for loss.item()), eager without that sync, the same stock kernels captured as a HIP graph
    (policy_trainer.GraphedTrainStep), and the hand-written step (native_train.NativeTrainStep, csrc/train_kernels.hip; no
    autograd, no MIOpen) -- with its exact-fp32 MFMA kernels (`native_fp32`: AZX_TRAIN_FWD / _BWD / _WGRAD=fp32) and as it
    ships, on the scaled split-f16 arithmetic (`native`).  Each mode runs in a child process of its own."""
    import subprocess
    out = {"what": "policy_trainer.supervised_step(train=True): %dx%d on %dx%d, SGD(momentum 0.9, weight decay 1e-4), "
                   "batch 128 collated from a 20000-row HBM ring, fp32; every mode in a fresh process"
                   % (args.blocks, args.chans, args.board, args.board), "batch": 128}
    for mode in TRAIN_MODES:
        cmd = [sys.executable, os.path.abspath(__file__), "--train-step-only", mode, "--board", str(args.board),
               "--blocks", str(args.blocks), "--chans", str(args.chans)]
        env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local_rank)))
        if mode == "native_fp32":
            env.update(AZX_TRAIN_FWD="fp32", AZX_TRAIN_BWD="fp32", AZX_TRAIN_WGRAD="fp32")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out[mode] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-400:]}
    # the same step at BASELINE configs[4]'s network (19x256 on 13x13), where it is throughput-bound: stock kernels
    # captured as a HIP graph against the hand-written wide step
    wide = {"what": "19x256 on 13x13 (BASELINE configs[4]'s network), batch 128, SGD(momentum 0.9, weight decay 1e-4), rows "
                    "collated from the HBM ring; each mode in a fresh process", "batch": 128}
    for mode in ("hip_graph", "native"):
        cmd = [sys.executable, os.path.abspath(__file__), "--train-step-only", mode, "--board", "13", "--blocks", "19",
               "--chans", "256", "--train-steps", "20"]
        env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local_rank)))
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        wide[mode] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-400:]}
    if all("steps_per_sec" in wide[m] for m in ("hip_graph", "native")):
        wide["speedup_native_vs_hip_graph"] = wide["native"]["steps_per_sec"] / wide["hip_graph"]["steps_per_sec"]
    out["wide"] = wide
    if all("steps_per_sec" in out[m] for m in TRAIN_MODES):
        out["speedup_native_vs_hip_graph"] = out["native"]["steps_per_sec"] / out["hip_graph"]["steps_per_sec"]
        out["speedup_split_f16_vs_fp32_kernels"] = out["native"]["steps_per_sec"] / out["native_fp32"]["steps_per_sec"]
        out["speedup_native_vs_reference_style_eager"] = out["native"]["steps_per_sec"] / out["eager"]["steps_per_sec"]
        out["speedup_native_vs_eager_nosync"] = out["native"]["steps_per_sec"] / out["eager_nosync"]["steps_per_sec"]
        out["speedup_hip_graph_vs_eager_nosync"] = out["hip_graph"]["steps_per_sec"] / out["eager_nosync"]["steps_per_sec"]
    return out


def run_train_loop(args, local_rank):
    """The trainer's loop on this one GPU (policy_trainer.py:82-90: a step, then `consume(batch / oversampling)`), at the
    headline's configuration, two ways -- the refills played inline (the deterministic mode) and self-play running beside
    the steps (azalea_amd/play_ahead.py: process_pool.py:29-47's in-flight games) -- each in a process of its own
    (tools/bench_train_loop.py).  steps/s of both, their ratio, and how much of the overlapped run the trainer waited."""
    import subprocess
    out = {"what": "NativeTrainStep (batch 128) + DeviceReplayBuffer.consume(12.8) per step, %dx%d on %dx%d, %d games, %d sims"
                   % (args.blocks, args.chans, args.board, args.board, args.games, args.sims),
           "games": args.games, "simulations": args.sims, "steps": args.loop_steps}
    tool = os.path.join(ROOT, "tools", "bench_train_loop.py")
    env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local_rank)))
    for mode in ("inline", "overlapped"):
        cmd = [sys.executable, tool, "--one", mode, "--steps", str(args.loop_steps), "--games", str(args.games), "--sims", str(args.sims)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode or not lines:
            out["error"] = "%s: %s" % (mode, (r.stderr or r.stdout)[-300:])
            return out
        out[mode] = json.loads(lines[-1])
    # over whole refill cycles (tools/bench_train_loop.py: a window of N steps holds 4 or 5 refills by chance)
    out["inline_steps_per_sec"] = out["inline"].get("steps_per_sec_whole_cycles", out["inline"]["steps_per_sec"])
    out["overlapped_steps_per_sec"] = out["overlapped"].get("steps_per_sec_whole_cycles", out["overlapped"]["steps_per_sec"])
    out["overlap_speedup"] = out["overlapped_steps_per_sec"] / out["inline_steps_per_sec"]
    out["overlapped_wait_share"] = out["overlapped"].get("wait_share")
    pa = out["overlapped"].get("play_ahead") or {}
    out["overlapped_weight_syncs"] = pa.get("weight_syncs")
    out["overlapped_max_backlog_rows"] = pa.get("max_backlog_rows")
    # staleness bound (azalea_amd/play_ahead.py): weight_sync_steps + (ahead_rows + one harvest) / rows consumed per step
    if pa.get("max_backlog_rows"):
        out["max_staleness_steps"] = pa.get("weight_sync_steps", 50) + pa["max_backlog_rows"] / 12.8
    return out


def run_api(args, rank, world, local_rank, start, torch):
    """The product surface for a whole game length: Player.read (parallel_player.py:24-28 -- what
    ReplayBuffer.consume calls, replay_buffer.py:121-132) on a device-policy Player at the headline's
    configuration, driven until every slot has played `--api-moves` moves since the pool was transplanted
    (transplanted games only carry the rows played since; after a full game length every game handed over is
    a whole one).  rows/s over the last 60 moves beside the plies/s the engine played in that window."""
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    n = args.board
    dev = "cuda:%d" % local_rank
    cfg = dict(device=dev, network="HexNetwork", board_size=n, num_blocks=args.blocks, base_chans=args.chans,
               simulations=args.sims, search_batch_size=args.batch, exploration_coef=0.5, exploration_depth=15,
               exploration_noise_alpha=0.03, exploration_noise_scale=args.noise_scale,
               exploration_temperature=1.0, seed=1)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device=dev)
    player = Player(None, [agent], n_games=args.games, gather=False)
    E = player.device_engine()
    player.prepare_device_engine(E)
    if start is not None:
        E.reset(moves=start)
    tot = {"plies": 0, "dev": 0.0}
    real_play = E.play

    def play(*a, **k):
        rows, st = real_play(*a, **k)
        tot["plies"] += st["plies"]
        tot["dev"] += st["seconds"]
        return rows, st
    E.play = play
    chunk = max(1, 8 * args.games)                 # ~8 engine moves' worth of rows per read in steady state
    t0 = time.perf_counter()
    reads = [(0.0, 0, 0, 0.0)]                      # wall, rows handed over, plies played, device seconds
    rows_total, metrics_keys = 0, set()
    while tot["plies"] < args.api_moves * args.games:
        frame, metrics = player.read(chunk)
        rows_total += len(frame)
        metrics_keys |= set(metrics)
        reads.append((time.perf_counter() - t0, rows_total, tot["plies"], tot["dev"]))
    player.stop()
    wall, _, plies, dev_s = reads[-1]
    # the window: from the first read that ends at or after (api_moves - 60) moves to the last one
    lo = next(i for i, r in enumerate(reads) if r[2] >= max(0, args.api_moves - 60) * args.games or i == len(reads) - 1)
    lo = min(lo, len(reads) - 2)
    w0, w1 = reads[lo], reads[-1]
    dt = w1[0] - w0[0]
    return {"surface": "Player.read (parallel_player.py:24-28) on the engine, gather off, %d games" % args.games,
            "moves_since_transplant": plies / args.games, "reads": len(reads) - 1, "rows": rows_total, "seconds": wall,
            "rows_per_sec": (w1[1] - w0[1]) / dt, "plies_per_sec": (w1[2] - w0[2]) / dt,
            "rows_over_plies": (w1[1] - w0[1]) / max(1, w1[2] - w0[2]),
            "window": "moves %.0f..%.0f after the transplant" % (w0[2] / args.games, w1[2] / args.games),
            "rows_per_sec_whole_run": rows_total / wall,
            "host_overhead_frac": (wall - dev_s) / wall, "metric_keys": sorted(metrics_keys)}


def rank_report(dist, torch, local_rank, st, elapsed):
    """N > 1: what every rank measured by itself, gathered as objects -- its device's identity (so that the line can say on
    how many DISTINCT GPUs the job ran), its own sims/s over its own clock.  None at N = 1."""
    if dist is None:
        return None
    p = torch.cuda.get_device_properties(local_rank)
    ident = "%s|%s|%s" % (os.uname().nodename, getattr(p, "uuid", None), getattr(p, "pci_bus_id", local_rank))
    mine = {"rank": dist.get_rank(), "device": ident, "local_rank": local_rank, "sims_per_sec": st["selects"] / elapsed,
            "seconds": elapsed}
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, mine)
    return parts


def world_fields(per_rank, value, exchange, ref_n1, backend):
    """Flat scalars of an N > 1 line (they go into `roofline`, which the driver's record keeps): distinct devices, the
    per-rank spread, the replay all-gather's rate and -- when the N = 1 value is given -- the weak-scaling efficiency."""
    w = len(per_rank)
    rates = [r["sims_per_sec"] for r in per_rank]
    out = {"world_ranks": w, "world_backend": backend, "world_distinct_devices": len({r["device"] for r in per_rank}),
           "world_rank_sims_per_sec_min": min(rates), "world_rank_sims_per_sec_max": max(rates),
           "world_rank_sims_per_sec_sum": sum(rates), "world_slowest_over_fastest": min(rates) / max(rates)}
    # RCCL carried the job only if every rank had a GPU of its own
    out["rccl_ranks"] = out["world_distinct_devices"] if backend == "nccl" else 0
    if exchange is not None and exchange.get("allgather_seconds"):
        out["replay_allgather_gbs"] = exchange["bytes_gathered"] / exchange["allgather_seconds"] / 1e9
        out["replay_allgather_ms"] = 1e3 * exchange["allgather_seconds"]
        out["replay_allgather_bytes"] = exchange["bytes_gathered"]
    if ref_n1:
        out["world_ref_n1_sims_per_sec"] = ref_n1
        out["weak_scaling_eff"] = value / (w * ref_n1)
    return out


def reduce_over_ranks(dist, torch, elapsed, sums):
    if dist is None:
        return elapsed, sums
    cdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor(sums, device=cdev, dtype=torch.float64)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in s.tolist()]


SUM_KEYS = ("selects", "games", "plies", "evals", "positions", "sum_game_length", "game_errors")


def throughput_fields(st_sums, elapsed, steps):
    selects, games, plies, evals, positions, glen, errors = st_sums
    out = {"value": selects / elapsed, "unit": "sims/s",
           "games_per_sec": games / elapsed, "plies_per_sec": plies / elapsed,
           "mean_game_length": (glen / games) if games else None,
           "games_per_sec_steady": (plies / elapsed) / (glen / games) if games and glen else None,
           "ms_per_step": 1e3 * elapsed / steps,
           "plies": plies, "games_finished": games, "game_errors": errors, "replay_rows": positions,
           "evals": evals, "elapsed_s": elapsed}
    return out


def tree_roofline(st, args, steps, warmup, kernels=None):
    b = model_bytes(st)
    achieved = b / st["mcts_seconds"] / 1e9 if st["mcts_seconds"] > 0 else 0.0
    kl = max(1, st.get("mcts_kernel_launches") or st["mcts_launches"])
    persistent = kl < st["mcts_launches"]
    roof = {
        "kernel": ("k_play<2> (select+expand+backup, move draw and game step of every game; "
                   "one persistent launch for all %d timed moves)" % steps) if persistent else
                  "k_mcts<2, FAST> (select+expand+backup, one launch per move)",
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "bytes_per_launch": b / kl,
        "avg_launch_ms": 1e3 * st["mcts_seconds"] / kl,
        "launches": kl,
        "moves_per_launch": st["mcts_launches"] / kl,
        "ms_per_move": 1e3 * st["mcts_seconds"] / max(1, st["mcts_launches"]),
        "bytes_per_sim": b / max(1, st["selects"]),
        "mean_depth": st["sum_depth"] / max(1, st["selects"]),
        "note": ("algorithmic bytes follow SURVEY 8(d)'s six-array reference model; the kernel keeps the root level "
                 "on chip, so its measured HBM traffic is below the model"),
        "limiter": ("dependent-latency chain, not HBM bandwidth and not issue: 410 serial simulations per game-wave, each "
                    "a chain of ~1.84 dependent child-block loads, 4 waves per SIMD; VALU busy 17 %, SALU+VALU issue 32 % "
                    "(profiles/*_tree_pmc_counters.json; DESIGN 3.1)"),
    }
    key = [args.games, args.board, args.sims, args.batch, steps, warmup, args.noise_scale, args.desync, args.settle]
    roof["traffic"], src = pmc_traffic(PMC_FILES["tree"], key, kernels)
    roof["traffic_source"] = src
    if roof["traffic"]:
        # what the HBM actually delivered, beside the model fraction above (the model prices the reference's
        # six-array layout; the kernel moves fewer bytes and is latency-bound)
        roof["achieved_hbm_gbs"] = roof["traffic"] / (1e-3 * roof["avg_launch_ms"]) / 1e9
        roof["achieved_hbm_frac"] = roof["achieved_hbm_gbs"] / HBM_PEAK_GBS
        roof["frac_is"] = "model bytes (SURVEY 8(d)) / time / peak; achieved_hbm_frac = counter bytes / time / peak"
    return roof


def resnet_roofline(st, args, steps, warmup, kernels=None):
    # dominant kernels: the residual tower + heads, one launch pair per leaf batch, timed with HIP events on
    # the engine stream.  ALGORITHMIC flops (107 851 784 per evaluated position) over that
    # time, against the dense MFMA peak of the dtype the tower issues: f16 (the fp32 operands are carried as
    # hi+lo f16 pairs, 3 MFMAs per product).
    flops = st["evals"] * flop_per_position(args.board, args.blocks, args.chans)
    net_s = st["net_seconds"] if st["net_seconds"] > 0 else st["seconds"]
    achieved = flops / net_s / 1e12
    if args.chans == 64 and args.board <= 11:
        kern = "k_tower_f16x3_s16 + k_heads"
    elif args.chans % 128 == 0:
        kern = "k_conv_wide_f16x3_s16 x %d + k_heads" % (2 * args.blocks)
    else:
        kern = "tower + k_heads"
    roof = {
        "kernel": kern + " (%dx%d resnet forward of one leaf batch)" % (args.blocks, args.chans),
        "bound": "mfma", "achieved": achieved, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "frac": achieved / F16_MFMA_PEAK_TF, "traffic": None,
        "issued_mfma_tflops": 3.0 * achieved, "issued_frac": 3.0 * achieved / F16_MFMA_PEAK_TF,
        "vs_fp32_mfma_peak": achieved / F32_MFMA_PEAK_TF,
        "avg_launch_ms": 1e3 * net_s / max(1, st["net_launches"]), "launches": st["net_launches"],
        "positions_per_launch": st["evals"] / max(1, st["net_launches"]),
        "flop_per_launch": flops / max(1, st["net_launches"]),
        "net_share_of_step": net_s / st["seconds"] if st["seconds"] > 0 else None,
        "tree_share_of_step": st["mcts_seconds"] / st["seconds"] if st["seconds"] > 0 else None,
    }
    key = [args.games, args.board, args.sims, args.batch, args.blocks, args.chans, steps, warmup,
           args.noise_scale, args.desync, args.settle]
    wide = args.chans % 128 == 0
    if wide:      # per-forward counters of the wide tower depend on the batch and the network only (tools/prof.sh)
        key = [args.games, args.board, args.batch, args.blocks, args.chans, "per forward"]
    roof["traffic"], src = pmc_traffic(PMC_FILES["config5" if wide else "resnet"], key, kernels)
    roof["traffic_source"] = src
    if roof["traffic"]:
        heads = PMC_EXTRA.get(PMC_FILES["config5" if wide else "resnet"], {}).get("heads_hbm_bytes_per_launch")
        what = ("one leaf-batch forward: the stem + %d per-layer launches" % (2 * args.blocks)) if wide \
            else "one k_tower_f16x3_s16 launch"
        if heads is not None:     # the pair `achieved` is timed over: tower + heads
            roof["traffic_tower"], roof["traffic_heads"] = roof["traffic"], heads
            roof["traffic"] = roof["traffic"] + heads
            roof["traffic_scope"] = "HBM bytes of %s + the k_heads launch behind it" % what
        else:
            roof["traffic_scope"] = "HBM bytes of %s (k_heads not counted)" % what
    return roof


def replay_exchange(E, dist, torch, local_rank, exchange_plies=0):
    """One replay refill the way a DeviceReplayBuffer shared by all ranks takes it: the rows this
    rank harvested are packed into fixed-size records on the device, all-gathered over RCCL and appended to
    the rank's HBM ring.  The bench's harvest queue wraps around, so the most recent rows are used."""
    import numpy as np
    world = dist.get_world_size() if dist is not None else 1
    dev = torch.device("cuda", local_rank)
    if exchange_plies > 0:
        # exactly `exchange_plies` more moves of every slot (at most 2N - 1: a slot then finishes at most one game,
        # the queue bound below is never reached early), so the set of games handed over is a function of the
        # pool alone -- the same for any number of ranks (tests/test_gpu_distributed.py)
        assert exchange_plies <= 2 * E.n - 1
        rows, _ = E.play_device(E.G * E.cells, max_plies=exchange_plies)
    else:
        rows, _ = E.play_device(1, max_plies=40)    # a short top-up through the Player.read path: whole games, queued
    rb = E.record_bytes
    E.replay_create(max(1, rows) * world + 1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    rec = torch.empty((rows, rb), dtype=torch.uint8, device=dev)
    if rows:
        E.rows_pack(0, rows, rec.data_ptr())
    t_pack = time.perf_counter() - t0
    t_gather = 0.0
    parts, counts = [rec], [rows]
    if dist is not None:
        from azalea_amd import distributed as azdist
        t1 = time.perf_counter()
        parts, counts = azdist.all_gather_records(rec)
        torch.cuda.synchronize(dev)
        t_gather = time.perf_counter() - t1
    t2 = time.perf_counter()
    for p, c in zip(parts, counts):
        if c:
            E.replay_put_records(c, p.data_ptr())
    t_put = time.perf_counter() - t2
    total = int(np.sum(counts))
    uids = [p[:, :8].contiguous().view(torch.int64).flatten() for p, c in zip(parts, counts) if c]
    uids = torch.cat(uids).unique().cpu().numpy() if uids else np.zeros(0, np.int64)
    return {"ranks": world, "rows_per_rank": counts, "record_bytes": rb, "bytes_gathered": total * rb,
            "games": int(len(uids)), "game_uid_sum": int(uids.sum()), "game_uid_xor": int(np.bitwise_xor.reduce(uids)) if len(uids) else 0,
            "game_uids": uids.tolist() if len(uids) <= 256 else None,
            "pack_seconds": t_pack, "allgather_seconds": t_gather, "ring_put_seconds": t_put,
            "path": "k_rows_pack -> all_gather(counts) + all_gather(records, device tensors) -> k_records_put"}


# the reference and the C port side by side ON ONE MACHINE (the survey / build container, 8 cores, where the reference
# can run): BASELINE.md section 2 for the reference, `oracle.bench_selfplay(11, 400, 10, T, T, net, max_plies=4,
# start_max=92)` for the port.  The GPU box's host has more, slower-per-thread cores; the reference cannot travel there.
SAME_CONTAINER = {"reference_per_core_8": 376.4, "port_per_core_8": 380.8, "reference_single_thread": 437.0,
                  "port_single_thread": 546.1, "unit": "sims/s",
                  "what": "11x11, 400 sims, 6x64, build container (8 cores): the port is not slower than the reference "
                          "per core on the same machine; the lower per_core on the GPU box is that host at 64 threads"}


def pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def box_calibration(torch):
    """How fast THIS box is, by a yardstick that is not this repo's code: a hipBLASLt f16 GEMM (8192^3, torch.matmul),
    best of three bursts of 30.  Leases of this pool differ by ~14 % in it (1 150 vs 1 310 TFLOP/s an hour apart) and
    the tower's launch time moves with it one to one (tests/test_gpu_perf_floor.py), so a headline read beside
    `box.gemm_f16_8192_tflops` can be compared across runs.  Outside every timed region."""
    n, iters = 8192, 30
    a = torch.randn(n, n, device="cuda", dtype=torch.float16)
    b = torch.randn(n, n, device="cuda", dtype=torch.float16)
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
    return {"gemm_f16_8192_tflops": best, "usual_on_this_pool": 1310.0, "relative": best / 1310.0,
            "what": "library f16 GEMM on this box (not this repo's code): a yardstick for box-to-box differences"}


def _scalar(v):
    return v is not None and isinstance(v, (bool, int, float, str))


def put_flat(dst, prefix, src, keys):
    """Copy the scalar entries `keys` of `src` into `dst` under `prefix` (`key` or `(key, new_name)`)."""
    if not isinstance(src, dict):
        return
    for k in keys:
        k, name = (k, k) if isinstance(k, str) else k
        if _scalar(src.get(k)):
            dst[prefix + name] = src[k]


def flatten_legs(line):
    """The driver's record keeps the SCALAR entries of `roofline`, `cpu_baseline` and `config` and nothing nested inside them
    (BENCH_r05.parsed: a nested `legs` dict was dropped), and every other nested dict of the line by name only.  So every
    leg is repeated here as flat scalar keys of those two dicts -- `tree_*` (configs[1]), `c5_*` (configs[4]'s shape on one
    GPU), `train_*`, `loop_*`, `api_*`, `box_*` -- enough that the record alone recomputes each fraction:
        tree_frac        = tree_bytes_per_launch / (tree_avg_launch_ms * 1e-3) / 1e9 / tree_peak_gbs
        tree_achieved_hbm_frac = tree_traffic / (tree_avg_launch_ms * 1e-3) / 1e9 / tree_peak_gbs
        c5_frac          = c5_flop_per_launch / (c5_avg_launch_ms * 1e-3) / 1e12 / c5_peak_tflops
        train_native_frac      = train_flop_per_step / (train_native_step_only_ms * 1e-3) / 1e12 / 2500
        train_wide_frac        = train_wide_flop_per_step / (train_wide_native_step_only_ms * 1e-3) / 1e12 / 2500
    A leg that failed leaves `<leg>_error`."""
    roof, cpu = line.get("roofline"), line.get("cpu_baseline")
    if not isinstance(roof, dict):
        return
    TK = (("value", "sims_per_sec"), "games_per_sec", "plies_per_sec", "ms_per_step", "steps", "warmup", "elapsed_s", "evals")
    for name, pre in (("tree", "tree_"), ("config5", "c5_")):
        leg = line.get(name)
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            roof[pre + "error"] = str(leg["error"])[:120]
            continue
        put_flat(roof, pre, leg, TK)
        r = leg.get("roofline") or {}
        put_flat(roof, pre, r, ("frac", "achieved", "traffic", "traffic_tower", "traffic_heads", "issued_frac", "avg_launch_ms",
                                "launches", "positions_per_launch", "flop_per_launch", "bytes_per_launch", "moves_per_launch",
                                "ms_per_move", "bytes_per_sim", "mean_depth", "achieved_hbm_gbs", "achieved_hbm_frac",
                                "net_share_of_step", ("unit", "achieved_unit"), ("bound", "bound")))
        if _scalar(r.get("peak")):
            roof[pre + ("peak_gbs" if r.get("bound") == "hbm" else "peak_tflops")] = r["peak"]
        put_flat(roof, pre, leg.get("config"), (("games_per_gpu", "games"), "board", "simulations"))
    ts = line.get("train_step")
    if isinstance(ts, dict):
        if ts.get("error") is not None:
            roof["train_error"] = str(ts["error"])[:120]
        put_flat(roof, "train_", ts, ("batch", "speedup_native_vs_hip_graph"))
        for mode, tag in (("eager", "eager"), ("eager_nosync", "eager_nosync"), ("hip_graph", "hipgraph"),
                          ("native_fp32", "native_fp32"), ("native", "native")):
            m = ts.get(mode)
            if isinstance(m, dict) and "error" in m:
                roof["train_%s_error" % tag] = str(m["error"])[:120]
            put_flat(roof, "train_%s_" % tag, m, (("ms_per_step", "ms"), "step_only_ms", ("frac_of_f16_mfma_peak", "frac"),
                                                  ("issued_frac_of_f16_mfma_peak", "issued_frac")))
        put_flat(roof, "train_", ts.get("native"), ("flop_per_step",))
        w = ts.get("wide")
        if isinstance(w, dict):
            put_flat(roof, "train_wide_", w, ("batch", "speedup_native_vs_hip_graph"))
            for mode, tag in (("hip_graph", "hipgraph"), ("native", "native")):
                m = w.get(mode)
                if isinstance(m, dict) and "error" in m:
                    roof["train_wide_%s_error" % tag] = str(m["error"])[:120]
                put_flat(roof, "train_wide_%s_" % tag, m, (("ms_per_step", "ms"), "step_only_ms",
                                                           ("issued_frac_of_f16_mfma_peak", "issued_frac")))
            put_flat(roof, "train_wide_", w.get("native"), ("flop_per_step", ("frac_of_f16_mfma_peak", "frac")))
    put_flat(roof, "loop_", line.get("train_loop"), ("inline_steps_per_sec", "overlapped_steps_per_sec", "overlap_speedup",
                                                     "overlapped_wait_share", "overlapped_weight_syncs",
                                                     "overlapped_max_backlog_rows", "max_staleness_steps",
                                                     "games", "simulations", "steps", "error"))
    put_flat(roof, "api_", line.get("api"), ("rows_per_sec", "plies_per_sec", "rows_over_plies", "host_overhead_frac", "error"))
    put_flat(roof, "box_", line.get("box"), (("gemm_f16_8192_tflops", "gemm_tflops"), ("usual_on_this_pool", "usual_tflops"),
                                             "relative", "error"))
    put_flat(roof, "", line, ("games_per_sec", "plies_per_sec", "games_per_sec_steady", "mean_game_length", "evals", "elapsed_s"))
    if isinstance(cpu, dict):
        CK = ("value", "unit", "cores", "per_core", "seconds", "sample", "error", "port_vs_reference_per_core")
        for name, pre in (("tree", "tree_"), ("config5", "c5_")):
            c = (line.get(name) or {}).get("cpu_baseline") if isinstance(line.get(name), dict) else None
            put_flat(cpu, pre, c, CK)
            if isinstance(c, dict):
                put_flat(cpu, pre + "reference_", c.get("reference_shim"), ("per_core", "cores", "sims_per_s"))
                put_flat(cpu, pre + "reference_scaled_", c.get("reference_scaled"), ("value",))
        put_flat(cpu, "reference_", cpu.get("reference_shim"), ("per_core", "cores", "sims_per_s", "games_per_s"))
        put_flat(cpu, "reference_scaled_", cpu.get("reference_scaled"), ("value", "cores"))
        put_flat(cpu, "same_container_", cpu.get("same_container"), (("port_per_core_8", "port_per_core"),
                                                                     ("reference_per_core_8", "reference_per_core")))


def strip_to_driver_record(line):
    """What the driver's BENCH_rNN.parsed keeps of a line: the contract's top-level scalars, and of `roofline` /
    `cpu_baseline` / `config` only the scalar entries (strings cut at 128 characters).  tests/test_bench_helpers.py
    recomputes every leg from this."""
    TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
           "vs_baseline", "dtype", "data")
    out = {k: line[k] for k in TOP if k in line}
    for name in ("config", "roofline", "cpu_baseline"):
        d = line.get(name)
        if isinstance(d, dict):
            out[name] = {k: (v[:128] if isinstance(v, str) else v) for k, v in d.items()
                         if v is None or isinstance(v, (bool, int, float, str))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["selfplay", "tree", "resnet"], default="selfplay",
                    help="selfplay = configs[2] headline with the configs[1] tree-only numbers nested (default)")
    ap.add_argument("--tree-steps", type=int, default=130, help="timed moves of the nested tree-only run")
    ap.add_argument("--tree-warmup", type=int, default=20)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--board", type=int, default=11)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--chans", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0xBAD5EED5)
    ap.add_argument("--noise-scale", type=float, default=0.25)
    ap.add_argument("--desync", type=int, default=None,
                    help="start slot i from a seeded random legal position of 0..DESYNC plies (default: "
                         "about one game length, 92 on 11x11; 0 = every game from the empty board)")
    ap.add_argument("--settle", type=int, default=None,
                    help="untimed moves of uniform-prior self-play per slot that bring the pool to its steady "
                         "state before the warm-up (default 2 * board^2)")
    ap.add_argument("--nodes-per-game", type=int, default=0,
                    help="tree arena capacity per game (0 = engine default)")
    ap.add_argument("--c5-games", type=int, default=512, help="concurrent games of the nested configs[4]-shape leg")
    ap.add_argument("--c5-steps", type=int, default=3, help="timed moves of the nested configs[4]-shape leg (one warm-up move)")
    ap.add_argument("--no-config5", action="store_true", help="skip the nested configs[4]-shape leg")
    ap.add_argument("--api-moves", type=int, default=190,
                    help="engine moves the nested product-surface leg (Player.read) is driven for; 0 = skip")
    ap.add_argument("--no-train-step", action="store_true", help="skip the nested training-step leg (SURVEY 8(f).4)")
    ap.add_argument("--loop-steps", type=int, default=4000, help="timed steps of the nested train-loop leg (inline / overlapped); 0 = skip")
    ap.add_argument("--train-steps", type=int, default=200, help="internal: timed steps of a --train-step-only run")
    ap.add_argument("--train-step-only", choices=list(TRAIN_MODES), default=None,
                    help="internal: run ONE mode of the training-step leg in this process and print its JSON")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-replay-exchange", action="store_true")
    ap.add_argument("--ref-n1", type=float, default=None,
                    help="N > 1: the N = 1 value (sims/s) this job's weak-scaling efficiency is quoted against in `world_*`")
    ap.add_argument("--all-legs", action="store_true",
                    help="N > 1: also run the nested tree / configs[4]-shape legs on every rank (default: headline + exchange only)")
    ap.add_argument("--exchange-plies", type=int, default=0,
                    help="replay exchange after exactly this many further moves of every slot (default: a short top-up "
                         "until the first game finishes)")
    args = ap.parse_args()
    headline = "tree" if args.workload == "tree" else "resnet"
    if args.steps is None:
        args.steps = 130 if headline == "tree" else 20
    if args.warmup is None:
        args.warmup = 20 if headline == "tree" else 5
    if args.desync is None:
        args.desync = int(round(0.76 * args.board * args.board))     # mean self-play game length, random-init net
    if args.settle is None:
        args.settle = 2 * args.board * args.board if args.desync else 0

    if args.train_step_only:
        import torch
        print(json.dumps(train_step_mode(args, 0, torch, args.train_step_only)))
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    ndev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL; AZX_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a
        # box with fewer GPUs than ranks (ranks then share devices: a functional check, not a benchmark)
        backend = os.environ.get("AZX_BENCH_BACKEND", "nccl")
        if backend == "nccl" and ndev < world:
            raise SystemExit("bench.py --gpus %d needs %d GPUs (found %d)" % (world, world, ndev))
        local_rank = local_rank % max(1, ndev)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine is HIP-only (no CPU fallback)")

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- headline workload --------------------------------------------------------------------
    st, elapsed, ex = run_workload(headline, args, rank, world, local_rank, args.steps, args.warmup, sync, torch)
    per_rank = rank_report(dist, torch, local_rank, st, elapsed)      # before the max-over-ranks reduction
    elapsed, sums = reduce_over_ranks(dist, torch, elapsed, [float(st[k]) for k in SUM_KEYS])
    E = ex["engine"]
    exchange = None
    if headline == "resnet" and not args.no_replay_exchange:
        exchange = replay_exchange(E, dist, torch, local_rank, args.exchange_plies)
    E.close()

    selects_per_search = (args.sims // args.batch + 1) * args.batch
    common_cfg = {"games_per_gpu": args.games, "board": args.board, "simulations": args.sims,
                  "search_batch_size": args.batch, "c_puct": 0.5, "noise": "dirichlet(0.03) eps 0.25, device RNG",
                  "start": ("pool in steady state, slots de-synchronised (untimed set-up): seeded random legal positions "
                            "of 0..%d plies, then %d moves of uniform-prior self-play per slot (finished games "
                            "restart in place), the positions reached are the start; then %d warm-up moves"
                            % (args.desync, args.settle, args.warmup)) if args.desync
                           else "all games from the empty board (lock-step)",
                  "sharding": "games sharded across ranks by global game index, no data-path collective"}
    line = None
    if rank == 0:
        line = {"metric": "mcts_sims_per_sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic"}
        line.update(throughput_fields(sums, elapsed, args.steps))
        line["world"] = {"ranks": world, "backend": (dist.get_backend() if dist is not None else None),
                         "games_per_rank": args.games}
        if exchange is not None:
            line["world"]["rows_per_rank"] = exchange["rows_per_rank"]
        line["kernels"] = ex["kernels"]
        if headline == "tree":
            wl = ("BASELINE configs[1]: %d concurrent %dx%d Hex games per GPU, HIP movegen+MCTS kernels only, "
                  "uniform priors (no net), %d sims/move (%d select_leaf calls)"
                  % (args.games, args.board, args.board, args.sims, selects_per_search))
            line["roofline"] = tree_roofline(st, args, args.steps, args.warmup, ex["kernels"])
        else:
            name = ("BASELINE configs[2]" if (args.board, args.blocks, args.chans) == (11, 6, 64) else
                    "BASELINE configs[4] shape on one GPU" if (args.board, args.blocks, args.chans) == (13, 19, 256)
                    else "resnet self-play")
            wl = ("%s: %d concurrent %dx%d Hex games per GPU, %d sims/move (%d select_leaf calls), %dx%d resnet "
                  "forward on split-f16 MFMA (fp32-accurate), random-init weights"
                  % (name, args.games, args.board, args.board, args.sims, selects_per_search, args.blocks, args.chans))
            line["dtype"] = "f16x3 (fp32 operands split hi+lo f16, fp32 accumulate)"
            line["roofline"] = resnet_roofline(st, args, args.steps, args.warmup, ex["kernels"])
            if exchange is not None:
                line["replay_allgather"] = exchange
        line["config"] = dict(workload=wl, **common_cfg)
        if per_rank is not None:
            line["world"]["per_rank"] = per_rank
            line["roofline"].update(world_fields(per_rank, line["value"], exchange, args.ref_n1,
                                                 dist.get_backend() if dist is not None else None))
        if world == 1:
            try:
                line["box"] = box_calibration(torch)
            except Exception as exc:
                line["box"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, headline, ex["net_state"])
            except Exception as exc:  # the oracle is test infrastructure; report, don't fail the bench
                line["cpu_baseline"] = {"error": repr(exc)}

    # ---- nested tree-only sub-benchmark (configs[1]), measured the same way ---------------------
    nested = args.workload == "selfplay" and (world == 1 or args.all_legs)    # N > 1: eight ranks each settling three engines
    if nested:                                                                # is the slow part of a SCALE run
        st_t, el_t, ex_t = run_workload("tree", args, rank, world, local_rank, args.tree_steps, args.tree_warmup,
                                        sync, torch)
        el_t, sums_t = reduce_over_ranks(dist, torch, el_t, [float(st_t[k]) for k in SUM_KEYS])
        ex_t["engine"].close()
        if rank == 0:
            tree = {"metric": "mcts_sims_per_sec", "steps": args.tree_steps, "warmup": args.tree_warmup, "dtype": "f32"}
            tree.update(throughput_fields(sums_t, el_t, args.tree_steps))
            tree["config"] = {"workload": ("BASELINE configs[1]: %d concurrent %dx%d Hex games per GPU, HIP movegen+MCTS "
                                           "kernels only, uniform priors (no net), %d sims/move (%d select_leaf calls)"
                                           % (args.games, args.board, args.board, args.sims, selects_per_search)),
                              "start": common_cfg["start"].replace("%d warm-up" % args.warmup, "%d warm-up" % args.tree_warmup)}
            tree["roofline"] = tree_roofline(st_t, args, args.tree_steps, args.tree_warmup, ex_t["kernels"])
            tree["kernels"] = ex_t["kernels"]
            if world == 1 and not args.no_cpu_baseline:
                try:
                    tree["cpu_baseline"] = cpu_baseline(args, "tree")
                except Exception as exc:
                    tree["cpu_baseline"] = {"error": repr(exc)}
            line["tree"] = tree

    # ---- nested configs[4]-shape leg: 13x13, 19x256, 810 selects, one warm-up + `--c5-steps` timed moves --------------
    if nested and not args.no_config5 and (args.board, args.blocks, args.chans) == (11, 6, 64):
        a5 = config5_args(args)
        try:
            st5, el5, ex5 = run_workload("resnet", a5, rank, world, local_rank, args.c5_steps, 1, sync, torch)
        except Exception as exc:      # a nested leg must not take the headline down (one process: no peer is left waiting)
            if world > 1:
                raise
            st5 = None
            line["config5"] = {"error": repr(exc)}
        if st5 is not None:
            el5, sums5 = reduce_over_ranks(dist, torch, el5, [float(st5[k]) for k in SUM_KEYS])
            ex5["engine"].close()
        if rank == 0 and st5 is not None:
            c5 = {"metric": "mcts_sims_per_sec", "steps": args.c5_steps, "warmup": 1,
                  "dtype": "f16x3 (fp32 operands split hi+lo f16, fp32 accumulate)"}
            c5.update(throughput_fields(sums5, el5, args.c5_steps))
            sel5 = (a5.sims // a5.batch + 1) * a5.batch
            c5["config"] = {"workload": ("BASELINE configs[4] shape on one GPU: %d concurrent 13x13 Hex games, %d sims/move "
                                         "(%d select_leaf calls), 19x256 resnet forward on split-f16 MFMA, random-init "
                                         "weights" % (a5.games, a5.sims, sel5)),
                            "games_per_gpu": a5.games, "board": 13, "simulations": a5.sims, "search_batch_size": a5.batch,
                            "start": "steady-state pool as the headline's (0..%d plies, %d settle moves), 1 warm-up move"
                                     % (a5.desync, a5.settle)}
            c5["roofline"] = resnet_roofline(st5, a5, args.c5_steps, 1, ex5["kernels"])
            c5["kernels"] = ex5["kernels"]
            if world == 1 and not args.no_cpu_baseline:
                try:      # 19x256 on the host: ~0.2 s per position and thread -> one ply of a 20-sim search per thread
                    c5["cpu_baseline"] = cpu_baseline(a5, "resnet", ex5["net_state"], sims=20, max_plies=1)
                except Exception as exc:
                    c5["cpu_baseline"] = {"error": repr(exc)}
            line["config5"] = c5

    # ---- nested product-surface leg: Player.read for a whole game length -----------------------------------
    if args.workload == "selfplay" and args.api_moves > 0 and world == 1:
        try:
            api = run_api(args, rank, world, local_rank, ex.get("start"), torch)
            line["api"] = api
            line["rows_per_sec"] = api["rows_per_sec"]
        except Exception as exc:
            line["api"] = {"error": repr(exc)}

    # ---- nested leg beside the path: the reference's training step on the replay ring, eager and captured ----
    if args.workload == "selfplay" and not args.no_train_step and world == 1 and (args.board, args.blocks, args.chans) == (11, 6, 64):
        try:
            line["train_step"] = run_train_step(args, local_rank, torch)
        except Exception as exc:
            line["train_step"] = {"error": repr(exc)}

    # ---- nested leg beside the path: the trainer's loop, refills inline against self-play beside the steps ---------
    if (args.workload == "selfplay" and not args.no_train_step and args.loop_steps > 0 and world == 1
            and (args.board, args.blocks, args.chans) == (11, 6, 64)):
        try:
            line["train_loop"] = run_train_loop(args, local_rank)
        except Exception as exc:
            line["train_loop"] = {"error": repr(exc)}

    if rank == 0:
        flatten_legs(line)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
