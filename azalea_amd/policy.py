"""Policy: network + MCTS behind the reference's interface (azalea/policy.py:21-208).

The search runs on the GPU (libazx_hip.so).  This class keeps what the reference keeps on the
host: the numpy RandomState (Dirichlet noise rows and the final move draw are produced here and
handed to the engine, so a fixed seed reproduces the reference's games), the exploration
schedule, and the checkpoint schema.  One Policy drives one engine slot ("parity mode");
bulk self-play goes through azalea_amd.parallel_player.Player ("throughput mode").
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import engine as _eng
from .utils import import_and_get


class SearchTreeFull(Exception):
    """The per-game node arena overflowed (azalea/search_tree.py:21, :258-259)."""


def as_distribution(counts: np.ndarray, temperature: float = 1.0) -> np.ndarray:
    """Visit counts -> move distribution, same numpy arithmetic as search_tree.py:327-344."""
    counts = np.asarray(counts, np.float32)
    assert (counts >= 0).all()
    with np.errstate(divide="ignore"):
        log_pi = np.log(counts.clip(min=1))
    log_pi[counts == 0] = -np.inf
    if temperature:
        log_pi = log_pi / temperature
    else:
        log_pi[log_pi < log_pi.max()] = -np.inf
    log_pi = log_pi.astype(np.float64)
    return np.exp(log_pi - np.logaddexp.reduce(log_pi))


def create_network(network_type, board_size, num_blocks, base_chans):
    """policy.py:11-18; the reference's own class names resolve to the engine's module."""
    if network_type in ("HexNetwork", "azalea.network.HexNetwork", "azalea_amd.network.HexNetwork"):
        from .network import HexNetwork
        return HexNetwork(board_size=board_size, num_blocks=num_blocks, base_chans=base_chans)
    Net = import_and_get(network_type)
    return Net(board_size=board_size, num_blocks=num_blocks, base_chans=base_chans)


_SEARCH_KEYS = ("simulations", "search_batch_size", "exploration_coef", "exploration_depth",
                "exploration_noise_alpha", "exploration_noise_scale", "exploration_temperature")
_NET_KEYS = ("network_type", "board_size", "num_blocks", "base_chans")


class Policy:
    def __init__(self):
        # greedy and deterministic unless the caller turns these on (policy.py:27-31)
        self.settings = {"move_sampling": False, "move_exploration": False}
        self.rng = np.random.RandomState()
        self.seed()
        self.ply = 0
        self._engine = None
        self._engine_key = None
        self._engine_moves = None     # moves applied to the engine slot since its last reset
        self._weights_version = None
        self.keep_reference_arena = True   # never-free arena + moving root, like the reference

    # ---- construction / persistence (policy.py:36-63, :85-130, :181-208) ------------------
    def initialize(self, config):
        device = torch.device(config["device"])
        self.net = create_network(config["network"], config["board_size"], config["num_blocks"],
                                  config["base_chans"])
        self.net.to(device)
        self.net.eval()
        self.network_type = config["network"]
        for k in ("board_size", "num_blocks", "base_chans"):
            setattr(self, k, config[k])
        for k in _SEARCH_KEYS:
            setattr(self, k, config[k])
        if "seed" in config:
            self.seed(config["seed"])

    @property
    def net(self):
        try:
            return self._net
        except AttributeError:
            raise RuntimeError("Policy must be initialized or loaded before use")

    @net.setter
    def net(self, net):
        self._net = net
        self._weights_version = None

    def state_dict(self):
        state = {"net": self.net.state_dict(), "rng": self.rng.__getstate__()}
        for k in _NET_KEYS + _SEARCH_KEYS:
            state[k] = getattr(self, k)
        return state

    def load_state_dict(self, state):
        for k in _NET_KEYS:
            setattr(self, k, state[k])
        self.net = create_network(self.network_type, self.board_size, self.num_blocks, self.base_chans)
        self.net.load_state_dict(state["net"])
        for k in _SEARCH_KEYS:
            setattr(self, k, state[k])
        if "rng" in state:
            self.rng.__setstate__(state["rng"])

    @classmethod
    def load(cls, path: str, device: Optional[str] = None) -> "Policy":
        policy = cls()
        location = None
        if device:
            device = torch.device(device)
            location = device.type + (":%d" % (device.index or 0) if device.type == "cuda" else "")
        if path.startswith("s3://"):
            import smart_open   # optional dependency, as in the reference
            with smart_open.smart_open(path) as f:
                state = torch.load(f, map_location=location, weights_only=False)
        else:
            state = torch.load(path, map_location=location, weights_only=False)
        policy.load_state_dict(state["policy"])
        policy.net.eval()
        if device:
            policy.net.to(device)
        return policy

    # ---- game lifecycle ---------------------------------------------------------------------
    def reset(self):
        """Start a new game (policy.py:76-80): fresh search tree."""
        self.ply = 0
        self._engine_moves = None

    def seed(self, seed: Optional[int] = None) -> None:
        self.rng.seed(seed)

    # ---- engine plumbing --------------------------------------------------------------------
    def _uses_device_net(self):
        from .network import HexNetwork
        return isinstance(self._net, HexNetwork)

    def _get_engine(self, board_size):
        device = 0
        if self._uses_device_net() and self._net.device.type == "cuda":
            device = self._net.device.index or 0
        key = (board_size, self.simulations, self.search_batch_size, float(self.exploration_coef),
               self._uses_device_net(), getattr(self, "num_blocks", 0), getattr(self, "base_chans", 0),
               self.keep_reference_arena, device)          # a net moved to another GPU gets a new engine there
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            sel = (self.simulations // self.search_batch_size + 1) * self.search_batch_size
            cells = board_size * board_size
            cap = (sel + 1) * cells * (cells + 1) // 2 + 1024 if self.keep_reference_arena else 0
            self._engine = _eng.Engine(
                board_size=board_size, n_games=1, simulations=self.simulations,
                search_batch_size=self.search_batch_size, exploration_coef=self.exploration_coef,
                exploration_depth=self.exploration_depth, noise_alpha=self.exploration_noise_alpha,
                noise_scale=self.exploration_noise_scale, temperature=self.exploration_temperature,
                evaluator=_eng.EVAL_RESNET if self._uses_device_net() else _eng.EVAL_EXTERNAL,
                num_blocks=getattr(self, "num_blocks", 0) or 0,
                base_chans=getattr(self, "base_chans", 0) or 1,
                nodes_per_game=min(cap, 1 << 23),
                flags=_eng.FLAG_NO_COMPACT if self.keep_reference_arena else 0, device=device)
            self._engine_key = key
            self._engine_moves = None
            self._weights_version = None
        return self._engine

    def _sync_weights(self, eng):
        if not self._uses_device_net():
            return
        sd = self._net.state_dict()
        version = tuple(int(t._version) for t in sd.values()) + (id(self._net),
                                                                 getattr(self._net, "weight_updates_outside_autograd", 0))
        if version == self._weights_version:
            return
        tensors = {k: v for k, v in sd.items() if v.dtype == torch.float32}
        if self._net.device.type == "cuda":
            eng.set_weights({k: (v.contiguous().data_ptr(), v.numel()) for k, v in tensors.items()},
                            on_device=True)
        else:
            eng.set_weights({k: v.detach().cpu().numpy() for k, v in tensors.items()})
        self._weights_version = version

    def _sync_position(self, eng, game):
        hist = getattr(game, "move_history", None)
        if hist is None:
            hist = _moves_from_board(game.state.board)
        if self._engine_moves is None or list(self._engine_moves) != list(hist):
            eng.reset(moves=[list(hist)])       # "step to the unknown": fresh tree at this position
            self._engine_moves = list(hist)

    def _host_evaluator(self):
        """mcts.evaluate_batch's network half (mcts.py:202-215) for duck-typed networks."""
        net = self._net

        def evaluate(boards, lm, slot, k):
            kmax = int(k.max())
            batch = {"board": torch.tensor(boards), "legal_moves": torch.tensor(lm[:, :kmax])}
            dev = getattr(net, "device", torch.device("cpu"))
            if dev.type == "cuda":
                batch = {n: t.pin_memory().to(dev) for n, t in batch.items()}
            out = net.run(batch)
            value = out["value"].cpu().numpy()
            prior = np.exp(out["moves_logprob"].cpu().numpy())
            return value, prior
        return evaluate

    # ---- acting (policy.py:132-176) ---------------------------------------------------------
    def choose_action(self, game) -> Tuple[int, Dict[str, Any]]:
        state = game.state
        assert not state.result
        temperature = noise_scale = 0.0
        if self.settings["move_sampling"]:
            temperature = self.exploration_temperature
            if self.settings["move_exploration"]:
                noise_scale = self.exploration_noise_scale
        if self.ply >= self.exploration_depth:
            temperature = 0.0        # noise is NOT gated by depth (policy.py:142-149)

        eng = self._get_engine(state.board.shape[0])
        self._sync_weights(eng)
        self._sync_position(eng, game)
        k = len(state.legal_moves)
        noise = None
        if noise_scale:
            # one Dirichlet draw per select_leaf, in call order (mcts.py:126-131)
            alpha = np.full(k, self.exploration_noise_alpha)
            noise = np.stack([self.rng.dirichlet(alpha) for _ in range(eng.selects_per_search)])[None]
        if self._uses_device_net():
            try:
                if hasattr(self._net, "eval"):
                    self._net.eval()
            except Exception:
                pass
            eng.search(noise=noise, noise_scale=noise_scale)
        else:
            eng.search_external(self._host_evaluator(), noise=noise, noise_scale=noise_scale)
        if eng.get_status()[0]:
            raise SearchTreeFull("too many nodes")
        root = eng.get_root()
        assert int(root["k"][0]) == k
        visits = root["child_visits"][0, :k]
        probs = as_distribution(visits, temperature)
        value = root["root_value"][0] / root["root_visits"][0]          # float32 (search_tree.py:108)
        metrics = {"search_value": root["search_value"][0],
                   "search_root_width": np.sum(visits > 0),
                   "search_root_visits": np.mean(visits),
                   "search_root_children": len(visits),
                   "search_tree_nodes": int(eng.get_tree_nodes()[0])}
        move_id = np.argmax(self.rng.multinomial(1, probs))
        move = state.legal_moves[move_id]
        info = dict(prob=probs[move_id], value=value, moves=state.legal_moves, moves_prob=probs,
                    move_id=move_id, metrics=metrics)
        return move, info

    def execute_action(self, move: int, legal_moves: np.ndarray) -> None:
        """Own or opponent move: re-root the tree (search_tree.py:115-132)."""
        move_id = legal_moves.tolist().index(move)
        if self._engine is not None and self._engine_moves is not None:
            self._engine.advance(np.array([move_id], np.int32))
            self._engine_moves.append(int(move))
        self.ply += 1

    def tree_metrics(self):
        return {}


def _moves_from_board(board):
    """Any alternating X/O order of the stones reproduces a (non-terminal) position."""
    flat = np.asarray(board).ravel()
    xs = (np.flatnonzero(flat == 1) + 1).tolist()
    os_ = (np.flatnonzero(flat == 2) + 1).tolist()
    moves = []
    for i in range(len(xs)):
        moves.append(xs[i])
        if i < len(os_):
            moves.append(os_[i])
    return moves
