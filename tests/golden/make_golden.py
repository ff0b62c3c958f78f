#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=tools/refshim:/root/reference python tests/golden/make_golden.py

`tools/refshim` holds an identity-decorator `numba` stand-in so the reference's
pure-Python jitted bodies run unmodified.  Nothing from the reference is copied:
this script drives the reference's public functions and records inputs/outputs.

Fixtures:
  g1_movegen.npz   seeded random playouts through azalea.game.hex.HexGame
  g2_flip.npz      HexGame.flip_player_board_moves on random padded batches
  g3_forward_*.npz HexNetwork.forward (weights + inputs + value/logprob)
  g4_search_*.npz  SearchTree.search with stub networks: evaluation tape + tree dump
  g5_game_*.npz    play_game([AzaleaAgent(Policy(stub))]) full self-play traces
  g5r_game_11_6x64.npz  one self-play game with the REAL network (11x11, 40 sims, the seeded 6x64 net of
                   g3_forward_11_6x64.npz): the trace plus every Network.run call's inputs and outputs
  g6_collate.npz   prep.torch_batch_replays on a small ReplayDataFrame
  g7_replay.npz    ReplayBuffer put/consume FIFO states and one shuffled DataLoader epoch
  g10_tournament.npz evaluation.evaluate round robin of three stub-net agents (per-game outcomes)
  g9_train_step.npz  three supervised_step calls (policy_trainer.py:123-142) on a fixed batch with
                   SGD(lr 0.1, momentum 0.9, weight_decay 1e-4): losses, outputs, updated tensors
  g8_checkpoint.npz  Policy.load of the shipped models/hex11-20180712-3362.policy.pth: schema,
                   per-tensor digests, forward outputs on 48 positions
"""
import os
import sys
import zlib

import numpy as np

np.seterr(over="ignore")
import torch  # noqa: E402

import azalea  # noqa: E402,F401
from azalea import mcts as ref_mcts  # noqa: E402
from azalea import prep as ref_prep  # noqa: E402
from azalea.azalea_agent import AzaleaAgent  # noqa: E402
from azalea.fnv1a import fnv1a  # noqa: E402
from azalea.game.hex import HexGame  # noqa: E402
from azalea.network import HexNetwork  # noqa: E402
from azalea.play_game import play_game  # noqa: E402
from azalea.policy import Policy  # noqa: E402
from azalea.search_tree import SearchTree  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


# --------------------------------------------------------------------------- G1
def random_playout(n, rng):
    """Play uniformly random legal moves through the reference game until it ends."""
    g = HexGame(n)
    moves, results, nlegal, lhash = [], [], [], []
    while True:
        st = g.state
        if st.result:
            break
        lm = st.legal_moves
        nlegal.append(len(lm))
        lhash.append(zlib.crc32(lm.astype(np.int32).tobytes()))
        mv = int(lm[rng.randint(len(lm))])
        g.step(mv)
        moves.append(mv)
        results.append(g.state.result)
    return moves, results, nlegal, lhash, g.state.board.copy()


def make_g1():
    rng = np.random.RandomState(20240101)
    out = {}
    for n, count in ((11, 800), (13, 260), (5, 100), (3, 40)):      # >= 1 000 on the two BASELINE board sizes
        cells = n * n
        mv = np.zeros((count, cells), np.int16)
        res = np.zeros((count, cells), np.int8)
        nl = np.zeros((count, cells), np.int16)
        lh = np.zeros((count, cells), np.uint32)
        ln = np.zeros(count, np.int16)
        fb = np.zeros((count, n, n), np.int8)
        for i in range(count):
            m, r, k, h, b = random_playout(n, rng)
            ln[i] = len(m)
            mv[i, :len(m)] = m
            res[i, :len(m)] = r
            nl[i, :len(m)] = k
            lh[i, :len(m)] = h
            fb[i] = b
        out.update({"moves_%d" % n: mv, "result_%d" % n: res, "nlegal_%d" % n: nl,
                    "legalcrc_%d" % n: lh, "length_%d" % n: ln, "final_%d" % n: fb})
    save("g1_movegen.npz", **out)


# --------------------------------------------------------------------------- G2
def make_g2():
    rng = np.random.RandomState(7)
    out = {}
    for n in (5, 11, 13):
        b = rng.randint(0, 3, size=(16, n, n)).astype(np.int32)
        k = n * n
        moves = np.zeros((16, k), np.int32)
        for i in range(16):
            empt = np.flatnonzero(b[i].ravel() == 0).astype(np.int32) + 1
            # ragged: keep a random-length ascending prefix-subset, zero padded
            keep = empt[: rng.randint(0, len(empt) + 1)]
            moves[i, :len(keep)] = keep
        fb, fm = HexGame.flip_player_board_moves(b, moves)
        out.update({"board_%d" % n: b, "moves_%d" % n: moves,
                    "fboard_%d" % n: np.ascontiguousarray(fb).astype(np.int32),
                    "fmoves_%d" % n: fm.astype(np.int32)})
    save("g2_flip.npz", **out)


# --------------------------------------------------------------------------- G3
def positions_from_playouts(n, count, rng):
    """Network inputs as mcts.evaluate_batch builds them: non-terminal states, boards of
    second-player-to-move rows flipped to the first player's view, legal moves padded."""
    states = []
    while len(states) < count:
        g = HexGame(n)
        stop = rng.randint(0, n * n)
        for _ in range(stop):
            st = g.state
            if st.result:
                break
            lm = st.legal_moves
            g.step(int(lm[rng.randint(len(lm))]))
        st = g.state
        if not st.result:
            states.append(st)
    batch = ref_prep.batch_games(states)
    board = batch["board"].copy()
    lm = batch["legal_moves"].copy()
    flip = batch["color"] == 1
    board[flip], lm[flip] = HexGame.flip_player_board_moves(board[flip], lm[flip])
    return board.astype(np.int32), lm.astype(np.int32), batch["color"].astype(np.int32)


def build_net(n, blocks, chans, seed):
    torch.manual_seed(seed)
    net = HexNetwork(board_size=n, num_blocks=blocks, base_chans=chans)
    # non-trivial BatchNorm statistics / affine so BN folding is really exercised
    g = torch.Generator().manual_seed(seed + 1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.5 + 0.25)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 1.0 + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    net.eval()
    # modern torch: the reference's permute leaves a channels_last view that .view() rejects;
    # hand the stem an equivalent NCHW-contiguous tensor (same values) via a forward hook.
    net.encoder.register_forward_hook(
        lambda mod, inp, out: out.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1))
    return net


def make_g3():
    rng = np.random.RandomState(33)
    for tag, n, blocks, chans, count in (("11_6x64", 11, 6, 64, 96),
                                         ("13_2x32", 13, 2, 32, 32),
                                         ("5_1x8", 5, 1, 8, 32)):
        net = build_net(n, blocks, chans, seed=1000 + n)
        board, lm, color = positions_from_playouts(n, count, rng)
        with torch.no_grad():
            o = net.run({"board": torch.tensor(board), "legal_moves": torch.tensor(lm)})
        arrays = {"w:" + k: v.detach().numpy() for k, v in net.state_dict().items()}
        arrays.update(board=board, legal_moves=lm, color=color,
                      value=o["value"].numpy(), moves_logprob=o["moves_logprob"].numpy(),
                      cfg=np.array([n, blocks, chans], np.int32))
        save("g3_forward_%s.npz" % tag, **arrays)


# --------------------------------------------------------------------------- stub nets
class StubNet:
    """Duck-typed azalea Network: .device, .eval(), .run(batch) -> value, moves_logprob.

    mode 'uniform0'   : logits 0 (uniform priors), value 0
    mode 'uniformhash': logits 0, value = (fnv1a(board) & 0xffff)/32768 - 1
    mode 'hashprior'  : logits from a hash of (board, tile), value from the board hash
    Padding is masked with -99 and log_softmax taken over the padded row, as the real net does.
    """
    device = torch.device("cpu")

    def __init__(self, mode):
        self.mode = mode

    def eval(self):
        return self

    def run(self, batch, compute_loss=False):
        board = batch["board"].numpy().astype(np.int32)
        lm = batch["legal_moves"].numpy()
        B, K = lm.shape
        value = np.zeros(B, np.float32)
        logit = np.zeros((B, K), np.float32)
        for i in range(B):
            h = int(fnv1a(board[i].ravel()))
            if self.mode != "uniform0":
                value[i] = np.float32((h & 0xFFFF) / 32768.0 - 1.0)
            if self.mode == "hashprior":
                for j in range(K):
                    t = int(lm[i, j])
                    if t:
                        x = (h ^ (t * 2654435761)) & 0xFFFFFFFF
                        x = (x * 2246822519) & 0xFFFFFFFF
                        logit[i, j] = np.float32(((x >> 13) & 0xFF) / 64.0)
        logit_t = torch.tensor(logit)
        logit_t.masked_fill_(torch.tensor(lm == 0), -99)
        lp = torch.log_softmax(logit_t, dim=1)
        return dict(value=torch.tensor(value), moves_logprob=lp)


class Tape:
    """Records every mcts.evaluate_batch result in call order (the 'evaluation tape')."""

    def __init__(self):
        self.values, self.nch, self.priors, self.boards, self.colors = [], [], [], [], []
        self._orig = ref_mcts.evaluate_batch

    def __enter__(self):
        def wrapped(game, net, states, rng):
            value, num_children, prior = self._orig(game, net, states, rng)
            for i, st in enumerate(states):
                self.values.append(np.float32(value[i]))
                self.nch.append(int(num_children[i]))
                self.priors.append(np.asarray(prior[i, :num_children[i]], np.float32).copy())
                self.boards.append(st.board.astype(np.int8).copy())
                self.colors.append(int(st.color))
            return value, num_children, prior
        ref_mcts.evaluate_batch = wrapped
        return self

    def __exit__(self, *exc):
        ref_mcts.evaluate_batch = self._orig

    def arrays(self, prefix=""):
        off = np.zeros(len(self.priors) + 1, np.int64)
        off[1:] = np.cumsum([len(p) for p in self.priors])
        flat = np.concatenate(self.priors) if self.priors else np.zeros(0, np.float32)
        return {prefix + "tape_value": np.array(self.values, np.float32),
                prefix + "tape_nch": np.array(self.nch, np.int32),
                prefix + "tape_prior": flat.astype(np.float32),
                prefix + "tape_off": off,
                prefix + "tape_board": np.array(self.boards, np.int8),
                prefix + "tape_color": np.array(self.colors, np.int8)}


def dump_tree(tree, prefix=""):
    n = tree.num_nodes
    return {prefix + "num_nodes": np.int64(n), prefix + "root_id": np.int64(tree.root_id),
            prefix + "parent": tree.parent[:n].copy(),
            prefix + "first_child": tree.first_child[:n].copy(),
            prefix + "num_children": tree.num_children[:n].copy(),
            prefix + "num_visits": tree.num_visits[:n].copy(),
            prefix + "total_value": tree.total_value[:n].copy(),
            prefix + "prior_prob": tree.prior_prob[:n].copy()}


class NoiseRecorder:
    """Wraps a RandomState so the dirichlet draws the search consumed can be stored."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.noise = []

    def dirichlet(self, alpha):
        x = self.rs.dirichlet(alpha)
        self.noise.append(x.copy())
        return x

    def multinomial(self, *a, **k):
        return self.rs.multinomial(*a, **k)


def play_prefix(n, nplies, seed):
    rng = np.random.RandomState(seed)
    g = HexGame(n)
    played = []
    for _ in range(nplies):
        st = g.state
        if st.result:
            break
        mv = int(st.legal_moves[rng.randint(len(st.legal_moves))])
        g.step(mv)
        if g.state.result:  # keep the search root non-terminal: stop before the winning move
            g = HexGame(n)
            for m in played:
                g.step(m)
            break
        played.append(mv)
    return g, np.array(played, np.int32)


def make_g4():
    # (tag, n, prefix plies, prefix seed, sims, batch, c_puct, mode, noise eps, alpha, rng seed, follow-up moves)
    cases = [
        ("a_11_empty_s40_u0", 11, 0, 0, 40, 10, 0.5, "uniform0", 0.0, 0.03, 0, 0),
        ("b_11_empty_s400_uh", 11, 0, 0, 400, 10, 0.5, "uniformhash", 0.0, 0.03, 0, 0),
        ("c_11_p30_s400_hp", 11, 30, 5, 400, 10, 0.75, "hashprior", 0.0, 0.03, 0, 0),
        ("d_11_p100_s400_uh", 11, 100, 6, 400, 10, 0.5, "uniformhash", 0.0, 0.03, 0, 0),
        ("e_11_p90_s40_hp", 11, 90, 8, 40, 10, 0.5, "hashprior", 0.0, 0.03, 0, 0),
        ("f_5_p12_s100_uh", 5, 12, 9, 100, 10, 0.5, "uniformhash", 0.0, 0.03, 0, 0),
        ("g_13_empty_s40_hp", 13, 0, 0, 40, 10, 0.5, "hashprior", 0.0, 0.03, 0, 0),
        ("h_11_empty_s40_noise", 11, 0, 0, 40, 10, 0.5, "hashprior", 0.25, 0.03, 1234, 0),
        ("i_11_p40_s60_noise_b7", 11, 40, 11, 60, 7, 0.75, "uniformhash", 0.25, 0.3, 99, 0),
        ("j_11_p20_s100_moves", 11, 20, 12, 100, 10, 0.5, "hashprior", 0.0, 0.03, 0, 3),
        ("k_5_p15_s100_moves", 5, 15, 13, 100, 10, 0.5, "uniformhash", 0.0, 0.03, 0, 3),
    ]
    for (tag, n, npre, pseed, sims, bs, c, mode, eps, alpha, rseed, follow) in cases:
        game, prefix = play_prefix(n, npre, pseed)
        net = StubNet(mode)
        tree = SearchTree()
        rng = NoiseRecorder(rseed)
        arrays = dict(cfg_n=np.int32(n), cfg_sims=np.int32(sims), cfg_batch=np.int32(bs),
                      cfg_c=np.float64(c), cfg_eps=np.float64(eps), cfg_alpha=np.float64(alpha),
                      cfg_seed=np.int64(rseed), prefix_moves=prefix,
                      mode=np.array(mode), follow=np.int32(follow))
        for step in range(follow + 1):
            pre = "s%d_" % step
            n_noise_before = len(rng.noise)
            with Tape() as tape:
                probs, value, metrics = tree.search(
                    game, net, num_simulations=sims, temperature=1.0,
                    exploration_coef=c, exploration_noise_scale=eps,
                    exploration_noise_alpha=alpha, batch_size=bs, rng=rng)
            arrays.update(tape.arrays(pre))
            arrays.update(dump_tree(tree, pre))
            arrays[pre + "probs"] = probs.astype(np.float64)
            arrays[pre + "value"] = np.float32(value)
            arrays[pre + "search_value"] = np.float64(metrics["search_value"])
            arrays[pre + "root_board"] = game.state.board.astype(np.int8)
            arrays[pre + "legal_moves"] = game.state.legal_moves.astype(np.int32)
            noise = rng.noise[n_noise_before:]
            if noise:
                arrays[pre + "noise"] = np.array(noise, np.float64)
            if step < follow:
                # deterministic follow-up: most visited root child (lowest index on ties)
                stats = tree.root.move_stats
                move_id = int(np.argmax(stats.num_visits))
                lm = game.state.legal_moves
                arrays[pre + "move_id"] = np.int32(move_id)
                tree.move(move_id)
                game.step(int(lm[move_id]))
                if game.state.result:
                    arrays["follow"] = np.int32(step)
                    break
        save("g4_search_%s.npz" % tag, **arrays)


# --------------------------------------------------------------------------- G5 / G6
def make_policy(mode, n, sims, bs, c, depth, alpha, eps, temp):
    p = Policy()
    p.net = StubNet(mode)
    p.network_type = "stub"
    p.board_size = n
    p.num_blocks = 0
    p.base_chans = 0
    p.simulations = sims
    p.search_batch_size = bs
    p.exploration_coef = c
    p.exploration_depth = depth
    p.exploration_noise_alpha = alpha
    p.exploration_noise_scale = eps
    p.exploration_temperature = temp
    return p


def make_g5_g6():
    frames = {}
    cases = [
        ("a_7_s20", 7, 20, 10, 0.5, 5, 0.03, 0.25, 1.0, "hashprior", 4242, True, True),
        ("b_11_s40", 11, 40, 10, 0.5, 15, 0.03, 0.25, 1.0, "uniformhash", 777, True, True),
        ("c_5_s30_greedy", 5, 30, 10, 0.75, 3, 0.3, 0.25, 1.0, "hashprior", 31337, False, False),
        ("d_9_s50_nonoise", 9, 50, 10, 0.5, 8, 0.03, 0.25, 0.5, "hashprior", 2718, True, False),
    ]
    for (tag, n, sims, bs, c, depth, alpha, eps, temp, mode, seed, sampling, explore) in cases:
        policy = make_policy(mode, n, sims, bs, c, depth, alpha, eps, temp)
        policy.settings["move_sampling"] = sampling
        policy.settings["move_exploration"] = explore
        agent = AzaleaAgent(lambda n=n: HexGame(n), policy=policy, device="cpu")
        agent.seed(seed)
        result, frame, metrics = play_game([agent], collect_data=True)
        P = len(frame)
        K = n * n
        board = np.zeros((P, n, n), np.int8)
        color = np.zeros(P, np.int8)
        nlegal = np.zeros(P, np.int16)
        lmoves = np.zeros((P, K), np.int16)
        mprob = np.zeros((P, K), np.float32)
        for i in range(P):
            st = frame.state[i]
            k = len(st.legal_moves)
            board[i] = st.board
            color[i] = st.color
            nlegal[i] = k
            lmoves[i, :k] = st.legal_moves
            mprob[i, :k] = frame.moves_prob[i]
            assert st.result == 0 and frame.moves_prob[i].dtype == np.float32
        mnames = sorted(k for k in metrics if k != "seconds_per_game")
        save("g5_game_%s.npz" % tag,
             cfg_n=np.int32(n), cfg_sims=np.int32(sims), cfg_batch=np.int32(bs),
             cfg_c=np.float64(c), cfg_depth=np.int32(depth), cfg_alpha=np.float64(alpha),
             cfg_eps=np.float64(eps), cfg_temp=np.float64(temp), mode=np.array(mode),
             cfg_seed=np.int64(seed), cfg_sampling=np.bool_(sampling), cfg_explore=np.bool_(explore),
             result=np.int32(result), board=board, color=color, nlegal=nlegal,
             legal_moves=lmoves, moves_prob=mprob, reward=np.array(frame.reward, np.float32),
             metric_names=np.array(mnames), metric_values=np.array([float(metrics[k]) for k in mnames]))
        frames[tag] = frame

    # G6: the reference collate over a ragged slice of one recorded game
    frame = frames["a_7_s20"]
    idx = [0, 3, 7, len(frame) - 1, len(frame) // 2]
    recs = [frame[i] for i in idx]
    tb = ref_prep.torch_batch_replays(recs)
    save("g6_collate.npz", idx=np.array(idx, np.int32),
         **{"out_" + k: v.numpy() for k, v in tb.items()},
         **{"dtype_" + k: np.array(str(v.numpy().dtype)) for k, v in tb.items()})
    make_g7(frames["b_11_s40"])


# --------------------------------------------------------------------------- G5 with the real network
class RunRecorder:
    """Records every Network.run call the search makes (mcts.py:202-206): the flipped input boards, the
    padded legal-move lists and the value / moves_logprob outputs, row by row in call order."""

    def __init__(self, net):
        self.net = net
        self.calls, self.boards, self.moves, self.values, self.logprobs = [], [], [], [], []
        self._orig = net.run

    def __enter__(self):
        def wrapped(batch, *a, **k):
            out = self._orig(batch, *a, **k)
            board = batch["board"].cpu().numpy()
            lm = batch["legal_moves"].cpu().numpy()
            value = out["value"].cpu().numpy()
            lp = out["moves_logprob"].cpu().numpy()
            self.calls.append(len(board))
            for i in range(len(board)):
                kk = int((lm[i] > 0).sum())
                assert (lm[i, :kk] > 0).all()
                self.boards.append(board[i].astype(np.int8))
                self.moves.append(lm[i, :kk].astype(np.int16))
                self.values.append(np.float32(value[i]))
                self.logprobs.append(lp[i, :kk].astype(np.float32))
            return out
        self.net.run = wrapped
        return self

    def __exit__(self, *exc):
        self.net.run = self._orig


def make_g5_real():
    """BASELINE configs[0] in small: ONE self-play game on 11x11, 40 simulations per move, search batch 10,
    the seeded 6x64 network of G3 (weights live in g3_forward_11_6x64.npz), the trainer's search
    settings (hex11_train_config.yml:19-36) with move sampling and exploration on."""
    n, sims, bs, c, depth, alpha, eps, temp, seed = 11, 40, 10, 0.5, 15, 0.03, 0.25, 1.0, 20240517
    policy = make_policy("uniform0", n, sims, bs, c, depth, alpha, eps, temp)
    policy.net = build_net(n, 6, 64, seed=1000 + n)          # the G3 network
    policy.network_type, policy.num_blocks, policy.base_chans = "hex", 6, 64
    policy.settings["move_sampling"] = True
    policy.settings["move_exploration"] = True
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cpu")
    agent.seed(seed)
    with RunRecorder(policy.net) as rec, torch.no_grad():
        result, frame, metrics = play_game([agent], collect_data=True)
    P, K = len(frame), n * n
    board = np.zeros((P, n, n), np.int8)
    color = np.zeros(P, np.int8)
    nlegal = np.zeros(P, np.int16)
    lmoves = np.zeros((P, K), np.int16)
    mprob = np.zeros((P, K), np.float32)
    for i in range(P):
        st = frame.state[i]
        k = len(st.legal_moves)
        board[i], color[i], nlegal[i] = st.board, st.color, k
        lmoves[i, :k] = st.legal_moves
        mprob[i, :k] = frame.moves_prob[i]
    off = np.zeros(len(rec.moves) + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in rec.moves])
    mnames = sorted(k for k in metrics if k != "seconds_per_game")
    save("g5r_game_11_6x64.npz",
         cfg_n=np.int32(n), cfg_sims=np.int32(sims), cfg_batch=np.int32(bs), cfg_c=np.float64(c),
         cfg_depth=np.int32(depth), cfg_alpha=np.float64(alpha), cfg_eps=np.float64(eps),
         cfg_temp=np.float64(temp), cfg_seed=np.int64(seed), cfg_net=np.array("g3_forward_11_6x64.npz"),
         result=np.int32(result), board=board, color=color, nlegal=nlegal, legal_moves=lmoves,
         moves_prob=mprob, reward=np.array(frame.reward, np.float32),
         metric_names=np.array(mnames), metric_values=np.array([float(metrics[k]) for k in mnames]),
         run_calls=np.array(rec.calls, np.int32), run_off=off,
         run_board=np.array(rec.boards, np.int8), run_moves=np.concatenate(rec.moves),
         run_value=np.array(rec.values, np.float32), run_logprob=np.concatenate(rec.logprobs))


# --------------------------------------------------------------------------- G7
def collate_padded(recs, cells):
    """torch_batch_replays of recs, legal_moves / moves_prob widened to `cells` columns + the width"""
    tb = {k: v.numpy() for k, v in ref_prep.torch_batch_replays(recs).items()}
    k = tb["legal_moves"].shape[1]
    for name in ("legal_moves", "moves_prob"):
        wide = np.zeros((len(recs), cells), tb[name].dtype)
        wide[:, :k] = tb[name]
        tb[name] = wide
    return tb, k


def make_g7(frame):
    """ReplayBuffer (replay_buffer.py:107-149) driven through puts that append, wrap and exceed the
    capacity; consume() against a stub player; one epoch of the trainer's shuffled DataLoader
    (policy_trainer.py:51-56).  `frame` is the recorded 11x11 game of G5 case b."""
    from azalea.replay_buffer import ReplayBuffer, ReplayDataFrame
    from torch.utils.data import DataLoader
    cells = 121
    cap = 12
    assert len(frame) >= cap + 7 + 8 + 30, len(frame)
    out = {}
    # every row of the source frame, collated once: tests rebuild frames from it
    src, _ = collate_padded([frame[i] for i in range(len(frame))], cells)
    for k, v in src.items():
        out["src_" + k] = v
    buf = ReplayBuffer(frame[:cap])
    cuts = [(cap, cap + 7), (cap + 7, cap + 15), (cap + 15, cap + 45)]
    out["cap"] = np.int32(cap)
    out["cuts"] = np.array(cuts, np.int32)
    for j, (a, b) in enumerate(cuts):
        buf.put(frame[a:b])
        tb, _ = collate_padded([buf[i] for i in range(len(buf))], cells)
        for k, v in tb.items():
            out["put%d_%s" % (j, k)] = v
        out["put%d_write_idx" % j] = np.int64(buf.write_idx)
        out["put%d_fresh" % j] = np.float64(buf.fresh_counter)

    class StubPlayer:            # hands out the next rows of the frame, at least `size` of them
        def __init__(self):
            self.pos = 0
            self.asked = []
        def read(self, size):
            n = int(np.ceil(size)) + 1
            self.asked.append(float(size))
            rows = frame[self.pos:self.pos + n]
            self.pos += n
            return rows, {"games": 1.0}
    buf2 = ReplayBuffer(frame[:cap])
    pl = StubPlayer()
    log = []
    for step in range(12):
        buf2.consume(5 / 2, pl)          # batch_size / oversampling, policy_trainer.py:90
        log.append((buf2.write_idx, buf2.fresh_counter, pl.pos))
    out["consume_log"] = np.array(log, np.float64)
    out["consume_asked"] = np.array(pl.asked, np.float64)
    tb, _ = collate_padded([buf2[i] for i in range(len(buf2))], cells)
    for k, v in tb.items():
        out["consume_final_" + k] = v

    torch.manual_seed(1234)
    loader = DataLoader(buf, batch_size=5, shuffle=True, num_workers=0,
                        collate_fn=ref_prep.torch_batch_replays)
    widths = []
    for j, batch in enumerate(loader):
        widths.append(batch["legal_moves"].shape[1])
        for k, v in batch.items():
            out["epoch_b%d_%s" % (j, k)] = v.numpy()
    out["epoch_widths"] = np.array(widths, np.int32)
    save("g7_replay.npz", **out)


# --------------------------------------------------------------------------- G8
def make_g8():
    """The reference's own checkpoint through the reference's Policy.load (policy.py:181-208):
    what a compatible loader must recover."""
    import hashlib
    path = os.path.join(os.path.dirname(os.path.dirname(azalea.__file__)), "models",
                        "hex11-20180712-3362.policy.pth")
    real_load = torch.load
    torch.load = lambda *a, **k: real_load(*a, **{**k, "weights_only": False})   # torch >= 2.6 default
    try:
        policy = Policy.load(path, device="cpu")
    finally:
        torch.load = real_load
    net = policy.net
    net.eval()
    net.encoder.register_forward_hook(
        lambda mod, inp, out: out.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1))
    rng = np.random.RandomState(88)
    board, lm, color = positions_from_playouts(policy.board_size, 48, rng)
    with torch.no_grad():
        o = net.run({"board": torch.tensor(board), "legal_moves": torch.tensor(lm)})
    sd = net.state_dict()
    names = sorted(sd)
    digests = [hashlib.sha256(np.ascontiguousarray(sd[k].numpy()).tobytes()).hexdigest() for k in names]
    hyper = ["network_type", "board_size", "num_blocks", "base_chans", "simulations", "search_batch_size",
             "exploration_coef", "exploration_depth", "exploration_noise_alpha", "exploration_noise_scale",
             "exploration_temperature"]
    save("g8_checkpoint.npz", file_bytes=np.int64(os.path.getsize(path)),
         file_sha256=np.array(hashlib.sha256(open(path, "rb").read()).hexdigest()),
         tensor_names=np.array(names), tensor_sha256=np.array(digests),
         tensor_shapes=np.array([str(tuple(sd[k].shape)) for k in names]),
         hyper_names=np.array(hyper), hyper_values=np.array([str(getattr(policy, k)) for k in hyper]),
         board=board, legal_moves=lm, color=color,
         value=o["value"].numpy(), moves_logprob=o["moves_logprob"].numpy(),
         cfg=np.array([policy.board_size, policy.num_blocks, policy.base_chans], np.int32),
         **{"w:" + k: sd[k].numpy() for k in names})


# --------------------------------------------------------------------------- G9
def make_g9():
    """The trainer's inner step (policy_trainer.py:123-142 supervised_step, network.py:92-102 loss)
    on a fixed batch of the recorded 11x11 game: train-mode BatchNorm, SGD with momentum and weight
    decay as train() configures them (policy_trainer.py:57-60)."""
    from azalea.policy_trainer import supervised_step
    from azalea.replay_buffer import ReplayDataFrame
    z = np.load(os.path.join(OUT, "g7_replay.npz"))
    recs = []
    from azalea.game.hex import HexGameState
    from azalea.replay_buffer import ReplayRecord
    for i in range(24):
        lm = z["src_legal_moves"][i]
        k = int((lm > 0).sum())
        st = HexGameState(int(z["src_color"][i]), lm[:k].astype(np.int32), int(z["src_result"][i]),
                          z["src_board"][i].astype(np.int32))
        recs.append(ReplayRecord(st, z["src_moves_prob"][i, :k].astype(np.float32), np.float32(z["src_reward"][i])))
    torch.manual_seed(77)
    net = HexNetwork(board_size=11, num_blocks=2, base_chans=16)
    net.encoder.register_forward_hook(
        lambda mod, inp, out: out.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1))
    init = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    out = {"w0:" + k: v for k, v in init.items()}
    idx = [list(range(0, 16)), list(range(8, 24)), list(range(4, 20))]
    out["batch_idx"] = np.array(idx, np.int32)
    for step, ids in enumerate(idx):
        batch = ref_prep.torch_batch_replays([recs[i] for i in ids])
        o, loss = supervised_step(net, batch, train=True, optimizer=opt, device="cpu")
        out["step%d_loss" % step] = np.float64(loss)
        out["step%d_value_loss" % step] = np.float64(o["value_loss"])
        out["step%d_moves_loss" % step] = np.float64(o["moves_loss"])
        out["step%d_value" % step] = o["value"].numpy()
        out["step%d_moves_logprob" % step] = o["moves_logprob"].numpy()
    for k, v in net.state_dict().items():
        out["w3:" + k] = v.numpy()
    batch = ref_prep.torch_batch_replays([recs[i] for i in idx[0]])
    o, loss = supervised_step(net, batch, train=False, device="cpu")
    out["eval_loss"] = np.float64(loss)
    out["eval_value"] = o["value"].numpy()
    save("g9_train_step.npz", **out)


# --------------------------------------------------------------------------- G10
def make_g10():
    """evaluation.evaluate (evaluation.py:17-80): three agents, three rounds of the round robin,
    in-process pool.  Per game: pair, seed, coin-flipped order, outcome; and the final tallies."""
    from azalea import evaluation as ref_eval
    n = 5
    specs = [("hashprior", 20, 0.5, True, True), ("uniformhash", 30, 0.75, True, False),
             ("hashprior", 40, 1.0, False, False)]
    agents = []
    for mode, sims, c, sampling, explore in specs:
        policy = make_policy(mode, n, sims, 10, c, 4, 0.3, 0.25, 1.0)
        policy.settings["move_sampling"] = sampling
        policy.settings["move_exploration"] = explore
        agents.append(AzaleaAgent(lambda n=n: HexGame(n), policy=policy, device="cpu"))
    games = []
    real_worker = ref_eval.worker

    def logging_worker(pair, ags, seed):
        rng = np.random.RandomState(seed)
        order = int(rng.choice([-1, 1]))
        pair2, outcome = real_worker(pair, ags, seed)
        games.append((pair[0], pair[1], seed, order, int(outcome)))
        return pair2, outcome
    ref_eval.worker = logging_worker
    try:
        outcomes = ref_eval.evaluate(agents, 3, num_workers=0)
    finally:
        ref_eval.worker = real_worker
    pairs = sorted(outcomes)
    save("g10_tournament.npz", board=np.int32(n),
         spec_mode=np.array([s[0] for s in specs]), spec_sims=np.array([s[1] for s in specs], np.int32),
         spec_c=np.array([s[2] for s in specs], np.float64),
         spec_sampling=np.array([s[3] for s in specs]), spec_explore=np.array([s[4] for s in specs]),
         games=np.array(games, np.int64), pairs=np.array(pairs, np.int32),
         tallies=np.array([outcomes[p] for p in pairs], np.int32))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g5r", "g8", "g9", "g10"]
    if "g5r" in which:
        make_g5_real()
    if "g1" in which:
        make_g1()
    if "g2" in which:
        make_g2()
    if "g3" in which:
        make_g3()
    if "g4" in which:
        make_g4()
    if "g5" in which:
        make_g5_g6()
    if "g8" in which:
        make_g8()
    if "g9" in which:
        make_g9()
    if "g10" in which:
        make_g10()
