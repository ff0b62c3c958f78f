#!/usr/bin/env python3
"""bench.py -- self-play throughput of the HIP engine (games/s + MCTS sims/s, 11x11 Hex @ 400 sims).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tree|resnet]

A "step" is one engine move: every one of the `--games` concurrent games runs a full
400-simulation search ((400//10+1)*10 = 410 select_leaf calls, mcts.py:268), draws its move on
the device and advances; finished games restart in place (with the uniform evaluator all timed
moves run in one persistent launch, every game looping search -> move draw -> step on its own
wavefront: K steps = K moves of each game).  Inputs are synthetic (all games start
from the empty board, random-init weights for the resnet workload) and already resident in HBM
when the timed region starts.

Default workload = BASELINE.json configs[1]: 4096 concurrent 11x11 games, HIP movegen + MCTS
kernels only, uniform priors (no net) -- HBM-bound, priced against the SURVEY 8(d) byte model.
`--workload resnet` = configs[2] (6x64 resnet forward per leaf batch, MFMA-bound).

N>1 is launched by the driver with torch.distributed.run, one rank per GPU: games shard
across ranks with no data-path collective (weak scaling); the barrier + max-over-ranks timing
use RCCL.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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


def cpu_baseline(args, net_state=None):
    """The CPU oracle (oracle/, a C restatement pinned against the reference's golden vectors)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    if args.workload == "tree":
        games, max_plies, net = 2 * threads, 300, None
        sample = "%d whole games, uniform priors, %d sims/move, one game per thread" % (games, args.sims)
    else:
        games, max_plies = threads, 2
        net = orc.Net(args.board, args.blocks, args.chans, net_state)
        sample = "%d games x first %d plies, 6x64 resnet fp32 direct conv, %d sims/move" % (
            games, max_plies, args.sims)
    out = orc.bench_selfplay(args.board, args.sims, args.batch, games, threads, net=net,
                             max_plies=max_plies, seed=args.seed)
    sims_per_s = out["selects"] / out["seconds"]
    return {"value": sims_per_s, "unit": "sims/s", "cores": threads, "kind": "port",
            "sample": sample, "seconds": out["seconds"], "plies": out["plies"],
            "games_per_s": (out["games"] / out["seconds"]) if args.workload == "tree" else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["tree", "resnet"], default="tree")
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--board", type=int, default=11)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--chans", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0xBAD5EED5)
    ap.add_argument("--noise-scale", type=float, default=0.25)
    ap.add_argument("--nodes-per-game", type=int, default=0,
                    help="tree arena capacity per game (0 = engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 130 if args.workload == "tree" else 4
    if args.warmup is None:
        args.warmup = 20 if args.workload == "tree" else 1

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

    from azalea_amd import engine as eng
    evaluator = eng.EVAL_UNIFORM if args.workload == "tree" else eng.EVAL_RESNET
    # per-game seeds come from the global game index: rank r owns uids r*2^40 + ...
    E = eng.Engine(board_size=args.board, n_games=args.games, simulations=args.sims,
                   search_batch_size=args.batch, exploration_coef=0.5, exploration_depth=15,
                   noise_alpha=0.03, noise_scale=args.noise_scale, temperature=1.0, evaluator=evaluator,
                   num_blocks=args.blocks, base_chans=args.chans, device=local_rank,
                   nodes_per_game=args.nodes_per_game,
                   seed=args.seed + (rank << 40))
    net_state = None
    if args.workload == "resnet":
        from azalea_amd.network import HexNetwork
        torch.manual_seed(0)
        net = HexNetwork(board_size=args.board, num_blocks=args.blocks, base_chans=args.chans)
        net.eval().to("cuda:%d" % local_rank)
        sd = net.state_dict()
        E.set_weights({k: (v.data_ptr(), v.numel()) for k, v in sd.items()
                       if v.dtype == torch.float32}, on_device=True)
        net_state = {k: v.detach().cpu().numpy() for k, v in sd.items()}

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup:
        E.play_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    st = E.play_steps(args.steps)          # blocks until the engine stream has drained
    sync()
    elapsed = time.perf_counter() - t0

    sums = [float(st[k]) for k in ("selects", "games", "plies", "evals", "positions")]
    if dist is not None:
        cdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor(sums, device=cdev, dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        sums = [float(x) for x in s.tolist()]
    selects, games, plies, evals, positions = sums

    if rank == 0:
        sims_per_s = selects / elapsed
        line = {
            "metric": "mcts_sims_per_sec", "value": sims_per_s, "unit": "sims/s",
            "games_per_sec": games / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[1]: %d concurrent %dx%d Hex games per GPU, HIP movegen+MCTS "
                             "kernels only, uniform priors (no net), %d sims/move (%d select_leaf calls)"
                             if args.workload == "tree" else
                             ("BASELINE configs[2]" if (args.board, args.blocks, args.chans) == (11, 6, 64) else
                              "BASELINE configs[4] shape on one GPU" if (args.board, args.blocks, args.chans) == (13, 19, 256)
                              else "resnet self-play") +
                             ": %d concurrent %dx%d Hex games per GPU, %d sims/move (%d select_leaf calls), " +
                             "%dx%d resnet forward on split-f16 MFMA (fp32-accurate), random-init weights"
                             % (args.blocks, args.chans))
                            % (args.games, args.board, args.board, args.sims,
                               (args.sims // args.batch + 1) * args.batch),
                "games_per_gpu": args.games, "board": args.board, "simulations": args.sims,
                "search_batch_size": args.batch, "c_puct": 0.5, "noise": "dirichlet(0.03) eps 0.25, device RNG",
                "sharding": "games sharded across ranks, no data-path collective",
            },
            "plies": plies, "games_finished": games, "replay_rows": positions,
            "evals": evals, "elapsed_s": elapsed,
        }
        # roofline of the dominant kernel, rank 0's own launches (HIP events on the engine stream)
        if args.workload == "tree":
            b = model_bytes(st)
            achieved = b / st["mcts_seconds"] / 1e9 if st["mcts_seconds"] > 0 else 0.0
            kl = max(1, st.get("mcts_kernel_launches") or st["mcts_launches"])
            persistent = kl < st["mcts_launches"]
            line["roofline"] = {
                "kernel": ("k_play<2> (select+expand+backup, move draw and game step of every game; "
                           "one persistent launch for all %d timed moves)" % args.steps) if persistent else
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
            }
            # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
            # (tools/prof_pmc.sh: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 read correction)
            tpath = os.path.join(ROOT, "profiles", "r1n_tree_pmc_traffic.json")
            default_cmd = (args.games, args.board, args.sims, args.batch, args.steps, args.warmup,
                           args.noise_scale) == (4096, 11, 400, 10, 130, 20, 0.25)
            if default_cmd and os.path.exists(tpath):
                t = json.load(open(tpath))
                line["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                line["roofline"]["traffic_source"] = "profiles/r1n_tree_pmc_traffic.json (PMC, same command)"
        else:
            # dominant kernels: the residual tower + heads, one launch pair per leaf batch, timed with
            # HIP events on the engine stream.  ALGORITHMIC flops (107 851 784 per
            # evaluated position) over that time, against the dense MFMA peak of the dtype the tower
            # issues: f16 (the fp32 operands are carried as hi+lo f16 pairs, 3 MFMAs per product).
            flops = st["evals"] * flop_per_position(args.board, args.blocks, args.chans)
            net_s = st["net_seconds"] if st["net_seconds"] > 0 else st["seconds"]
            achieved = flops / net_s / 1e12
            line["dtype"] = "f16x3 (fp32 operands split hi+lo f16, fp32 accumulate)"
            line["roofline"] = {
                "kernel": (("k_tower_f16x3 + k_heads" if os.environ.get("AZX_TOWER_SHAPE") == "32" else
                            "k_tower_f16x3_s16 + k_heads") if args.chans == 64 and args.board <= 11 else
                           ("k_conv_wide_f16x3" if os.environ.get("AZX_TOWER_SHAPE") == "32" else "k_conv_wide_f16x3_s16") + " x %d + k_heads" % (2 * args.blocks) if args.chans % 128 == 0 else
                           "tower + k_heads") + " (%dx%d resnet forward of one leaf batch)" % (args.blocks, args.chans),
                "bound": "mfma", "achieved": achieved, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": achieved / F16_MFMA_PEAK_TF, "traffic": None,
                "issued_mfma_tflops": 3.0 * achieved, "issued_frac": 3.0 * achieved / F16_MFMA_PEAK_TF,
                "vs_fp32_mfma_peak": achieved / F32_MFMA_PEAK_TF,
                "avg_launch_ms": 1e3 * net_s / max(1, st["net_launches"]), "launches": st["net_launches"],
                "positions_per_launch": st["evals"] / max(1, st["net_launches"]),
                "net_share_of_step": net_s / st["seconds"] if st["seconds"] > 0 else None,
            }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, net_state)
            except Exception as exc:  # the oracle is test infrastructure; report, don't fail the bench
                line["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(line))
    E.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
