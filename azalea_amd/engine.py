"""Thin numpy front-end over the C ABI (include/azx.h): one Engine = one azx_engine handle.

Everything here is plumbing: arrays in, arrays out.  The search, the rules and the network
forward all run in libazx_hip.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (EVAL_EXTERNAL, EVAL_RESNET, EVAL_UNIFORM, EVAL_UNIFORM_HASH,  # noqa: F401
                   FLAG_NO_COMPACT, AzxError, Config, PlayStats, check)


def _p(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


class Engine:
    def __init__(self, board_size=11, n_games=1, simulations=400, search_batch_size=10,
                 exploration_coef=0.5, exploration_depth=15, noise_alpha=0.03, noise_scale=0.25,
                 temperature=1.0, evaluator=EVAL_UNIFORM, num_blocks=6, base_chans=64,
                 nodes_per_game=0, flags=0, device=0, seed=0xBAD5EED5,
                 game_index_stride=1, game_index_offset=0):
        self.L = _lib.lib()
        self.cfg = Config(board_size, n_games, simulations, search_batch_size,
                          float(np.float32(exploration_coef)), exploration_depth, noise_alpha,
                          noise_scale, temperature, evaluator, num_blocks, base_chans,
                          nodes_per_game, flags, device, seed, game_index_stride, game_index_offset)
        self.h = C.c_void_p()
        check(self.L.azx_create(C.byref(self.cfg), C.byref(self.h)))
        self.n = board_size
        self.cells = board_size * board_size
        self.G = n_games
        self.bs = search_batch_size
        self.num_batches = simulations // search_batch_size + 1          # mcts.py:268
        self.selects_per_search = self.num_batches * search_batch_size
        self._last_rows = 0

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.azx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- set-up -------------------------------------------------------------------------
    def set_prior_table(self, table):
        t = np.ascontiguousarray(table, np.float32)
        check(self.L.azx_set_prior_table(self.h, _p(t, C.c_float), t.size))

    def set_weights(self, tensors, on_device=False, sync=True):
        """tensors: {state_dict name: contiguous fp32 numpy array} or {name: (ptr, count)}.
        `on_device`: the pointers are device memory of this GPU (torch tensors).  The engine copies them through its
        own HIP runtime, which knows nothing of torch's streams, so whatever last wrote them -- an optimizer step
        still in flight, a graph replay -- is waited for here (`sync=False`: the caller has already waited for the
        writer, e.g. on an event; play_ahead.PlayAhead)."""
        if on_device and sync:
            import torch
            torch.cuda.synchronize(self.cfg.device)
        names, ptrs, counts, keep = [], [], [], []
        for name, t in tensors.items():
            if name.endswith("num_batches_tracked"):
                continue
            if isinstance(t, tuple):
                ptr, cnt = t
            else:
                a = np.ascontiguousarray(t, np.float32)
                keep.append(a)
                ptr, cnt = a.ctypes.data, a.size
            names.append(name.encode())
            ptrs.append(ptr)
            counts.append(cnt)
        n = len(names)
        c_names = (C.c_char_p * n)(*names)
        c_ptrs = (C.c_void_p * n)(*ptrs)
        c_counts = (C.c_int64 * n)(*counts)
        check(self.L.azx_set_weights(self.h, n, c_names, c_ptrs, c_counts, 1 if on_device else 0))

    def packed_weights(self):
        """{operand name: bytes} -- the packed weight buffers exactly as the kernels read them (azx_debug_weights)."""
        out, which = {}, 0
        while True:
            nbytes = C.c_int64(0)
            name = C.create_string_buffer(64)
            check(self.L.azx_debug_weights(self.h, which, None, 0, C.byref(nbytes), name, 64))
            if nbytes.value < 0:
                return out
            buf = np.empty(nbytes.value, np.uint8)
            check(self.L.azx_debug_weights(self.h, which, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(nbytes), name, 64))
            out[name.value.decode()] = buf
            which += 1

    def weights_digest(self):
        """sha256 over the packed operands: two engines search with the same network iff their digests agree."""
        import hashlib
        h = hashlib.sha256()
        for name, buf in sorted(self.packed_weights().items()):
            h.update(name.encode())
            h.update(buf.tobytes())
        return h.hexdigest()

    def reset(self, slots=None, moves=None):
        """Reset slots (all by default); `moves` = list of move lists replayed per slot."""
        s = None if slots is None else np.ascontiguousarray(slots, np.int32)
        ns = self.G if s is None else len(s)
        if moves is None:
            check(self.L.azx_reset(self.h, _p(s, C.c_int32), ns, None, None, 0))
            return
        stride = max(1, max(len(m) for m in moves))
        mv = np.zeros((ns, stride), np.int32)
        nm = np.zeros(ns, np.int32)
        for i, m in enumerate(moves):
            mv[i, :len(m)] = m
            nm[i] = len(m)
        check(self.L.azx_reset(self.h, _p(s, C.c_int32), ns, _p(mv, C.c_int32), _p(nm, C.c_int32), stride))

    # ---- search -------------------------------------------------------------------------
    def set_active(self, mask):
        m = np.ascontiguousarray(mask, np.int32)
        assert m.shape == (self.G,)
        check(self.L.azx_set_active(self.h, _p(m, C.c_int32)))

    def search(self, noise=None, noise_scale=0.0):
        """noise: float64 [G, n_select, stride] host Dirichlet rows, or None."""
        if noise is not None:
            nz = np.ascontiguousarray(noise, np.float64)
            assert nz.ndim == 3 and nz.shape[0] == self.G
            check(self.L.azx_search(self.h, _p(nz, C.c_double), nz.shape[1], nz.shape[2], noise_scale))
        else:
            check(self.L.azx_search(self.h, None, 0, 0, noise_scale))

    def search_begin(self, noise=None, noise_scale=0.0):
        n = C.c_int(0)
        if noise is not None:
            nz = np.ascontiguousarray(noise, np.float64)
            check(self.L.azx_search_begin(self.h, _p(nz, C.c_double), nz.shape[1], nz.shape[2],
                                          noise_scale, C.byref(n)))
        else:
            check(self.L.azx_search_begin(self.h, None, 0, 0, noise_scale, C.byref(n)))
        return n.value

    def search_step(self):
        n, done = C.c_int(0), C.c_int(0)
        check(self.L.azx_search_step(self.h, C.byref(n), C.byref(done)))
        return n.value, bool(done.value)

    def get_leaves(self):
        cap = self.G * self.bs
        boards = np.zeros((cap, self.n, self.n), np.int32)
        lm = np.zeros((cap, self.cells), np.int32)
        slot = np.zeros(cap, np.int32)
        k = np.zeros(cap, np.int32)
        n = C.c_int(0)
        check(self.L.azx_get_leaves(self.h, cap, _p(boards, C.c_int32), _p(lm, C.c_int32),
                                    _p(slot, C.c_int32), _p(k, C.c_int32), C.byref(n)))
        n = n.value
        return boards[:n], lm[:n], slot[:n], k[:n]

    def put_evals(self, value, prior):
        v = np.ascontiguousarray(value, np.float32)
        p = np.zeros((len(v), self.cells), np.float32)
        pr = np.asarray(prior, np.float32)
        if len(v):
            p[:, :pr.shape[1]] = pr
        check(self.L.azx_put_evals(self.h, len(v), _p(v, C.c_float), _p(p, C.c_float)))

    def get_evals(self):
        """Device-network results for the pending leaves (EVAL_RESNET + phase API)."""
        cap = self.G * self.bs
        v = np.zeros(cap, np.float32)
        p = np.zeros((cap, self.cells), np.float32)
        n = C.c_int(0)
        check(self.L.azx_get_evals(self.h, cap, _p(v, C.c_float), _p(p, C.c_float), C.byref(n)))
        return v[:n.value], p[:n.value]

    def search_recorded(self, noise=None, noise_scale=0.0):
        """EVAL_RESNET search driven phase by phase, returning the evaluation tape
        [(slot, k, value, prior[:k])] in mcts.evaluate_batch order."""
        tape = []
        n = self.search_begin(noise, noise_scale)
        while True:
            if n:
                b, lm, slot, k = self.get_leaves()
                v, p = self.get_evals()
                for i in range(len(k)):
                    tape.append((int(slot[i]), int(k[i]), np.float32(v[i]), p[i, :k[i]].copy()))
            n, done = self.search_step()
            if done:
                break
        return tape

    def search_external(self, evaluate, noise=None, noise_scale=0.0):
        """Drive one search with a host evaluator: evaluate(boards, legal_moves, slot, k) ->
        (value[n], prior[n, >=max k]).  Mirrors mcts.sample_paths' call order."""
        n = self.search_begin(noise, noise_scale)
        while True:
            if n:
                b, lm, slot, k = self.get_leaves()
                v, p = evaluate(b, lm, slot, k)
                self.put_evals(v, p)
            n, done = self.search_step()
            if done:
                break

    # ---- results ------------------------------------------------------------------------
    def get_root(self):
        G, Cn = self.G, self.cells
        out = dict(k=np.zeros(G, np.int32), legal_moves=np.zeros((G, Cn), np.int32),
                   child_visits=np.zeros((G, Cn), np.float32),
                   child_value=np.zeros((G, Cn), np.float32),
                   child_prior=np.zeros((G, Cn), np.float32), root_visits=np.zeros(G, np.float32),
                   root_value=np.zeros(G, np.float32), num_nodes=np.zeros(G, np.int32),
                   search_value=np.zeros(G, np.float32))
        check(self.L.azx_get_root(
            self.h, _p(out["k"], C.c_int32), _p(out["legal_moves"], C.c_int32),
            _p(out["child_visits"], C.c_float), _p(out["child_value"], C.c_float),
            _p(out["child_prior"], C.c_float), _p(out["root_visits"], C.c_float),
            _p(out["root_value"], C.c_float), _p(out["num_nodes"], C.c_int32),
            _p(out["search_value"], C.c_float)))
        return out

    def get_status(self):
        st = np.zeros(self.G, np.int32)
        check(self.L.azx_get_status(self.h, _p(st, C.c_int32)))
        return st

    def get_tree_nodes(self):
        """SearchTree.num_nodes as the reference counts it (never reclaimed; search_tree.py:112)."""
        nn = np.zeros(self.G, np.int32)
        check(self.L.azx_get_tree_nodes(self.h, _p(nn, C.c_int32)))
        return nn

    def get_games(self):
        G = self.G
        board = np.zeros((G, self.n, self.n), np.int32)
        color, result, ply = (np.zeros(G, np.int32) for _ in range(3))
        check(self.L.azx_get_games(self.h, _p(board, C.c_int32), _p(color, C.c_int32),
                                   _p(result, C.c_int32), _p(ply, C.c_int32)))
        return dict(board=board, color=color, result=result, ply=ply)

    def advance(self, move_ids):
        m = np.ascontiguousarray(move_ids, np.int32)
        assert m.shape == (self.G,)
        check(self.L.azx_advance(self.h, _p(m, C.c_int32)))

    def tree_dump(self, slot=0, cap=None):
        if cap is None:
            cap = int(self.get_root()["num_nodes"][slot]) + 8
        arrs = dict(parent=np.zeros(cap, np.int32), first_child=np.zeros(cap, np.int32),
                    num_children=np.zeros(cap, np.int32), num_visits=np.zeros(cap, np.float32),
                    total_value=np.zeros(cap, np.float32), prior_prob=np.zeros(cap, np.float32))
        nn, rid = C.c_int32(0), C.c_int32(0)
        check(self.L.azx_tree_dump(
            self.h, slot, cap, _p(arrs["parent"], C.c_int32), _p(arrs["first_child"], C.c_int32),
            _p(arrs["num_children"], C.c_int32), _p(arrs["num_visits"], C.c_float),
            _p(arrs["total_value"], C.c_float), _p(arrs["prior_prob"], C.c_float),
            C.byref(nn), C.byref(rid)))
        out = {k: v[:nn.value].copy() for k, v in arrs.items()}
        out["num_nodes"], out["root_id"] = nn.value, rid.value
        return out

    def forward(self, boards, legal_moves):
        b = np.ascontiguousarray(boards, np.int32)
        lm = np.ascontiguousarray(legal_moves, np.int32)
        B, K = lm.shape
        value = np.zeros(B, np.float32)
        logprob = np.zeros((B, K), np.float32)
        check(self.L.azx_forward(self.h, B, K, _p(b, C.c_int32), _p(lm, C.c_int32),
                                 _p(value, C.c_float), _p(logprob, C.c_float)))
        return value, logprob

    # ---- throughput mode ------------------------------------------------------------------
    def play(self, min_positions, max_plies=0):
        cap = int(min_positions) + self.G * self.cells
        board = np.zeros((cap, self.n, self.n), np.int32)
        color = np.zeros(cap, np.int32)
        nlegal = np.zeros(cap, np.int32)
        prob = np.zeros((cap, self.cells), np.float32)
        reward = np.zeros(cap, np.float32)
        uid = np.zeros(cap, np.int64)
        st = PlayStats()
        check(self.L.azx_play(self.h, int(min_positions), int(max_plies), cap,
                              _p(board, C.c_int32), _p(color, C.c_int32), _p(nlegal, C.c_int32),
                              _p(prob, C.c_float), _p(reward, C.c_float), _p(uid, C.c_int64),
                              C.byref(st)))
        n = st.positions
        self._last_rows = int(n)
        return dict(board=board[:n], color=color[:n], nlegal=nlegal[:n], moves_prob=prob[:n],
                    reward=reward[:n], game_uid=uid[:n]), st.as_dict()

    # azx_play_row_metrics columns -> the reference's per-ply metric names (search_tree.py:109-112, mcts.py:291,
    # policy.py:164); column 3 flags the first row of a game
    ROW_METRIC_COLUMNS = (("search_value", 0), ("search_root_width", 1), ("action_logprob", 2),
                          ("search_root_visits", 4), ("search_tree_nodes", 5), ("search_root_children", 6))

    def game_metric_sums(self, rows=None):
        """Sums over the games of the last play()/play_device()/replay_fill() call of each game's per-ply MEANS
        of the search metrics, by the reference's key names: what Player.read's metrics add up
        (play_game.py:73-76, parallel_player.py:50-51)."""
        m = self.play_row_metrics(rows)
        names = [k for k, _ in self.ROW_METRIC_COLUMNS]
        if len(m) == 0:
            return dict.fromkeys(names, 0.0)
        starts = np.flatnonzero(m[:, 3] > 0.5)
        if len(starts) == 0 or starts[0] != 0:
            starts = np.r_[0, starts]
        cols = [c for _, c in self.ROW_METRIC_COLUMNS]
        sums = np.add.reduceat(m[:, cols].astype(np.float64), starts, axis=0)
        lens = np.diff(np.r_[starts, len(m)])
        return dict(zip(names, (sums / lens[:, None]).sum(0).tolist()))

    def play_row_metrics(self, rows=None):
        """azx_play_row_metrics: [rows, AZX_ROW_METRICS] for the rows of the last play() / play_device() call
        (`rows`: how many that call returned, if known); columns: ROW_METRIC_COLUMNS, 3 = first row of a game."""
        cap = int(rows) if rows is not None else self._last_rows
        m = np.zeros((max(1, cap), _lib.ROW_METRICS), np.float32)
        n = C.c_int64(0)
        check(self.L.azx_play_row_metrics(self.h, max(1, cap), _p(m, C.c_float), C.byref(n)))
        return m[:n.value]

    def kernel_info(self):
        """azx_kernel_info: which kernels this engine launches (and the diagnostic switches it was created under)."""
        buf = C.create_string_buffer(1024)
        n = self.L.azx_kernel_info(self.h, buf, len(buf))
        if n < 0:
            check(n)
        return buf.value.decode()

    def debug_set_queue_cap(self, rows):
        """tests: bound the harvest queue of the following play calls (0 = no bound)."""
        check(self.L.azx_debug_set_queue_cap(self.h, int(rows)))

    def play_device(self, min_positions, max_plies=0):
        """azx_play_device: whole games until >= min_positions rows sit in the harvest queue (in HBM)."""
        st = PlayStats()
        rows = C.c_int64(0)
        check(self.L.azx_play_device(self.h, int(min_positions), int(max_plies), C.byref(rows), C.byref(st)))
        self._last_rows = int(rows.value)
        return rows.value, st.as_dict()

    def rows_pack(self, first, n, records_ptr):
        """queue rows [first, first+n) -> fixed-size records in the device buffer at records_ptr."""
        check(self.L.azx_rows_pack(self.h, int(first), int(n), C.c_void_p(int(records_ptr))))

    def replay_put_records(self, n, records_ptr):
        """n records from a device buffer into the replay ring (FIFO)."""
        check(self.L.azx_replay_put_records(self.h, int(n), C.c_void_p(int(records_ptr))))

    def replay_put_records_async(self, n, records_ptr, stream):
        """The same put enqueued on `stream` (a hipStream_t value) and not synchronised: the one engine call that may
        run on another host thread than a play in progress (include/azx.h)."""
        check(self.L.azx_replay_put_records_async(self.h, int(n), C.c_void_p(int(records_ptr)), C.c_void_p(int(stream))))

    def reserve_cus(self, cus_per_xcd):
        """Keep `cus_per_xcd` CUs of every XCD (rounded up to 4: one per shader engine) free of this engine's kernels;
        0 = use all.  Returns the CUs left free on the whole device."""
        out = C.c_int(0)
        check(self.L.azx_reserve_cus(self.h, int(cus_per_xcd), C.byref(out)))
        return out.value

    @property
    def record_bytes(self):
        return _lib.record_bytes(self.cells)

    def debug_counters(self):
        out = np.zeros(16, np.uint64)
        check(self.L.azx_debug_counters(self.h, _p(out, C.c_uint64)))
        return out

    # ---- device-resident replay ring (azx_replay_*) -------------------------------------------
    def replay_create(self, capacity):
        check(self.L.azx_replay_create(self.h, int(capacity)))

    def replay_state(self):
        v = [C.c_int64(0) for _ in range(3)]
        check(self.L.azx_replay_state(self.h, *(C.byref(x) for x in v)))
        return dict(capacity=v[0].value, size=v[1].value, write_idx=v[2].value)

    def replay_set_state(self, size, write_idx):
        check(self.L.azx_replay_set_state(self.h, int(size), int(write_idx)))

    def replay_put(self, board, color, nlegal, moves_prob, reward):
        """Host rows (azx_play layout: board i32[P,cells], moves_prob f32[P,cells] dense by child
        index) into the ring, FIFO."""
        P = len(reward)
        cells = self.n * self.n
        board = np.ascontiguousarray(board, np.int32).reshape(P, cells)
        prob = np.ascontiguousarray(moves_prob, np.float32).reshape(P, cells)
        color = np.ascontiguousarray(color, np.int32)
        nlegal = np.ascontiguousarray(nlegal, np.int32)
        reward = np.ascontiguousarray(reward, np.float32)
        check(self.L.azx_replay_put(self.h, P, _p(board, C.c_int32), _p(color, C.c_int32),
                                    _p(nlegal, C.c_int32), _p(prob, C.c_float), _p(reward, C.c_float)))

    def replay_fill(self, min_positions, max_plies=0):
        st = PlayStats()
        rows = C.c_int64(0)
        check(self.L.azx_replay_fill(self.h, int(min_positions), int(max_plies), C.byref(rows), C.byref(st)))
        self._last_rows = int(rows.value)
        return rows.value, st.as_dict()

    def replay_collate(self, indices, out):
        """indices: int64 host array; out: dict of device pointers (ints) color, legal_moves,
        result, board, moves_prob, reward.  Returns the batch's largest legal-move count."""
        idx = np.ascontiguousarray(indices, np.int64)
        mk = C.c_int32(0)
        check(self.L.azx_replay_collate(self.h, len(idx), _p(idx, C.c_int64),
                                        *(C.c_void_p(int(out[k])) for k in
                                          ("color", "legal_moves", "result", "board", "moves_prob", "reward")),
                                        C.byref(mk)))
        return mk.value

    def replay_set_mover_view(self, on: bool):
        """azx_replay_set_mover_view: the collates hand out the second player's rows as the search sees them."""
        check(self.L.azx_replay_set_mover_view(self.h, 1 if on else 0))

    def replay_collate_async(self, indices, out, stream):
        """replay_collate enqueued on `stream` (a hipStream_t as int) without synchronising and without max_k."""
        idx = np.ascontiguousarray(indices, np.int64)
        check(self.L.azx_replay_collate_async(self.h, len(idx), _p(idx, C.c_int64),
                                              *(C.c_void_p(int(out[k])) for k in
                                                ("color", "legal_moves", "result", "board", "moves_prob", "reward")),
                                              C.c_void_p(int(stream))))

    def debug_choose(self):
        """azx_debug_choose: (move_id[G], moves_prob[G, cells]) of the device move draw on the current roots."""
        mid = np.zeros(self.G, np.int32)
        prob = np.zeros((self.G, self.cells), np.float32)
        check(self.L.azx_debug_choose(self.h, _p(mid, C.c_int32), _p(prob, C.c_float)))
        return mid, prob

    def debug_counters_raw(self):
        out = np.zeros((self.G, 16), np.uint64)
        check(self.L.azx_debug_counters_raw(self.h, _p(out, C.c_uint64), self.G))
        return out

    def play_steps(self, plies):
        st = PlayStats()
        check(self.L.azx_play_steps(self.h, int(plies), C.byref(st)))
        return st.as_dict()


def hex_replay(board_size, moves, lengths, device=0):
    """azx_hex_replay: per-ply result / legal count / empties mask for many move lists."""
    mv = np.ascontiguousarray(moves, np.int32)
    ln = np.ascontiguousarray(lengths, np.int32)
    G, stride = mv.shape
    res = np.zeros((G, stride), np.int32)
    nl = np.zeros((G, stride), np.int32)
    em = np.zeros((G, stride, 4), np.uint64)
    fb = np.zeros((G, board_size, board_size), np.int32)
    check(_lib.lib().azx_hex_replay(device, board_size, G, _p(mv, C.c_int32), _p(ln, C.c_int32),
                                    stride, _p(res, C.c_int32), _p(nl, C.c_int32),
                                    _p(em, C.c_uint64), _p(fb, C.c_int32)))
    return res, nl, em, fb


def random_prefixes(board_size, indices, max_len, seed, device=0):
    """Seeded random legal move prefixes, one per global game index in `indices`, of uniformly drawn
    lengths in [0, max_len] (cut one move short of a win if the random stones happen to finish the game).
    Used to start a pool of concurrent games out of phase (bench.py: a pool that restarts finished games in
    place is, in steady state, spread over all plies -- not lined up on the empty board).  Game i's prefix
    depends on (seed, i) only, so it does not change with the number of ranks.  The rules run on the device
    (azx_hex_replay)."""
    idx = np.asarray(indices, np.int64)
    cells = board_size * board_size
    perms = np.empty((len(idx), cells), np.int32)
    want = np.empty(len(idx), np.int64)
    for j, i in enumerate(idx):
        rng = np.random.RandomState((int(seed) + 0x9E3779B1 * int(i)) % (2 ** 32))
        want[j] = rng.randint(0, max_len + 1)
        perms[j] = rng.permutation(cells) + 1
    res, _, _, _ = hex_replay(board_size, perms, np.full(len(idx), cells, np.int32), device=device)
    won = res != 0
    first_win = np.where(won.any(1), won.argmax(1), cells)      # stones on the board before the winning move
    length = np.minimum(want, first_win)
    return [perms[j, :length[j]].tolist() for j in range(len(idx))]


def selftest_arith(a, b, device=0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    sq, dv, mul = (np.zeros_like(a) for _ in range(3))
    check(_lib.lib().azx_selftest_arith(device, a.size, _p(a, C.c_float), _p(b, C.c_float),
                                        _p(sq, C.c_float), _p(dv, C.c_float), _p(mul, C.c_float)))
    return sq, dv, mul


def selftest_divide(num, den, device=0):
    """azx_selftest_divide: (num/den through the kernel's unscaled divide, sqrt-table entry of den)."""
    num = np.ascontiguousarray(num, np.float32)
    den = np.ascontiguousarray(den, np.float32)
    q, rt = np.zeros_like(num), np.zeros_like(num)
    check(_lib.lib().azx_selftest_divide(device, len(num), _p(num, C.c_float), _p(den, C.c_float),
                                         _p(q, C.c_float), _p(rt, C.c_float)))
    return q, rt


def selftest_dirichlet(alpha, k, n_rows, seed=1, device=0):
    out = np.zeros((n_rows, k), np.float32)
    check(_lib.lib().azx_selftest_dirichlet(device, float(alpha), k, n_rows, seed, _p(out, C.c_float)))
    return out
