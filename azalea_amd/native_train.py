"""The trainer's inner step on hand-written gfx950 kernels (csrc/train_kernels.hip).

`NativeTrainStep` stands where `policy_trainer.supervised_step(model, batch, train=True, optimizer=...)` stands in the
reference loop (azalea/policy_trainer.py:84-90, :123-142): one call = zero_grad + train-mode forward + the loss of
network.py:92-102 + backward + torch.optim.SGD's update, as ~40 kernels queued on torch's current stream (the filter
gradients on a second one; `AZX_TRAIN_GRAPH=1` captures the whole step as one HIP graph).  The 3x3 convolutions of all
three passes run on the split-f16 MFMA arithmetic of the self-play tower (hi + lo f16 operands, three products, fp32
accumulate: 22 significant bits) with every operand scaled per layer by a power of two -- the gradients from max |g_l|,
the filters from max |w|, the activations from their BatchNorm's bound -- so nothing depends on the magnitudes staying
inside the f16 range; `AZX_TRAIN_FWD=fp32`, `AZX_TRAIN_BWD=fp32`, `AZX_TRAIN_WGRAD=fp32` select the exact-fp32 MFMA
kernels per pass.  PyTorch keeps HOLDING everything: the kernels read and write the module's parameter tensors, its
BatchNorm buffers and the optimizer's momentum buffers in place, so checkpoints (policy_trainer.py:161-181), the StepLR
scheduler and Player's weight refresh see an ordinary module and optimizer.  No autograd, no MIOpen.
"""
import ctypes as C

import numpy as np
import torch
from torch import optim

from . import _lib


class _DeviceArray:
    """A device buffer of the native trainer seen through the CUDA array interface: torch.as_tensor aliases it."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def unsupported_reason(model, optimizer, device):
    """None when NativeTrainStep covers this (module, optimizer, device); otherwise why not -- policy_trainer.train
    picks the hand-written step by default where this returns None and the stock step elsewhere."""
    from .network import HexNetwork
    dev = torch.device(device)
    if dev.type != "cuda":
        return "NativeTrainStep needs a CUDA (ROCm) device"
    if not isinstance(model, HexNetwork):
        return "NativeTrainStep trains azalea_amd.network.HexNetwork"
    if not isinstance(optimizer, optim.SGD) or len(optimizer.param_groups) != 1:
        return "NativeTrainStep implements torch.optim.SGD with one parameter group"
    g = optimizer.param_groups[0]
    if g.get("dampening", 0) or g.get("nesterov", False) or g.get("maximize", False):
        return "NativeTrainStep: dampening / nesterov / maximize are not implemented"
    if {id(p) for p in g["params"]} != {id(p) for p in model.parameters()}:
        return "NativeTrainStep: the optimizer must hold exactly the module's parameters"
    chans = model.conv1.weight.shape[0]
    if (model.board_size, chans) not in SUPPORTED_SHAPES:
        return ("NativeTrainStep covers boards up to 11x11 with 16 / 32 / 64 channels, 12x12 / 13x13 with 64, and 3x3 .. 13x13 with 128 / 256; got %dx%d, %d channels"
                % (model.board_size, model.board_size, chans))
    return None


class _Shapes:
    """`(board_size, channels) in SUPPORTED_SHAPES`: what azx_train_create accepts (csrc/train_kernels.hip)."""

    def __contains__(self, key):
        n, c = key
        return (2 <= n <= 11 and c in (16, 32, 64)) or (3 <= n <= 13 and c in (128, 256)) or (12 <= n <= 13 and c == 64)


SUPPORTED_SHAPES = _Shapes()


class NativeTrainStep:
    """Same surface as policy_trainer.GraphedTrainStep: `step(batch)`, `step_from_ring(replaybuf, indices)`,
    `outputs(k)`, `.loss` (three device floats: total, value, moves).

    Covers HexNetwork under SGD (one parameter group, no dampening, no nesterov) on the shapes of SUPPORTED_SHAPES: 16 / 32 /
    64 channels up to 11x11, 64 / 128 / 256 up to 13x13; anything else raises ValueError -- use GraphedTrainStep or the
    eager step there."""

    def __init__(self, model, optimizer, batch_size: int, device):
        from .network import HexNetwork
        dev = torch.device(device)
        why = unsupported_reason(model, optimizer, dev)
        if why:
            raise ValueError(why)
        if dev.index is None:                      # "cuda" -> the current device, as torch resolves it
            dev = torch.device("cuda", torch.cuda.current_device())
        self.model, self.optimizer, self.B, self.device = model, optimizer, int(batch_size), dev
        n = model.board_size
        self.cells = n * n
        blocks, chans = len(model.resblocks), model.conv1.weight.shape[0]
        self._L = _lib.lib()
        cfg = _lib.TrainConfig(n, blocks, chans, self.B, dev.index or 0)
        self._h = C.c_void_p()
        _lib.check(self._L.azx_train_create(C.byref(cfg), C.byref(self._h)))
        ptrs = [C.c_void_p() for _ in range(7)]
        _lib.check(self._L.azx_train_inputs(self._h, *[C.byref(p) for p in ptrs[:4]]))
        _lib.check(self._L.azx_train_outputs(self._h, *[C.byref(p) for p in ptrs[4:]]))

        def view(p, shape, typestr):
            return torch.as_tensor(_DeviceArray(p.value, shape, typestr), device=dev)
        B, cells = self.B, self.cells
        self.board = view(ptrs[0], (B, n, n), "<i4")
        self.legal_moves = view(ptrs[1], (B, cells), "<i4")
        self.moves_prob = view(ptrs[2], (B, cells), "<f4")
        self.reward = view(ptrs[3], (B,), "<f4")
        self.loss = view(ptrs[4], (3,), "<f4")                 # total, value, moves
        self.out_value = view(ptrs[5], (B,), "<f4")
        self.out_logprob = view(ptrs[6], (B, cells), "<f4")
        self._color = torch.zeros(B, dtype=torch.int64, device=dev)
        self._result = torch.zeros(B, dtype=torch.int64, device=dev)
        self._bound_key = None
        self._quick_bound = None
        params = list(model.parameters())
        self._sentinels = (params[0], params[-1])
        self.steps = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.azx_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- binding ------------------------------------------------------------------------------
    def _tensors(self):
        """(name, tensor, momentum buffer or None) for every state_dict entry; creates missing momentum buffers as
        zeros (torch's SGD makes them on its first step as buf = d, which is what mu * 0 + d gives)."""
        params = dict(self.model.named_parameters())
        out = []
        for name, t in self.model.state_dict().items():
            mom = None
            if name in params:
                st = self.optimizer.state[params[name]]
                if st.get("momentum_buffer") is None:
                    st["momentum_buffer"] = torch.zeros_like(params[name], memory_format=torch.preserve_format)
                mom = st["momentum_buffer"]
            out.append((name, t, mom))
        return out

    def _key(self):
        params = dict(self.model.named_parameters())
        key = []
        for name, t in self.model.state_dict().items():
            mom = self.optimizer.state.get(params[name], {}).get("momentum_buffer") if name in params else None
            key.append((t.data_ptr(), None if mom is None else mom.data_ptr()))
        return key

    def _bind(self):
        names, ptrs, counts, moms = [], [], [], []
        for name, t, mom in self._tensors():
            if not t.is_contiguous() or t.device != self.device:
                raise ValueError("NativeTrainStep: '%s' must be a contiguous tensor on %s" % (name, self.device))
            if t.is_floating_point() and t.dtype != torch.float32:
                raise ValueError("NativeTrainStep: '%s' must be float32" % name)
            names.append(name.encode())
            ptrs.append(t.data_ptr())
            counts.append(t.numel())
            moms.append(None if mom is None else mom.data_ptr())
        n = len(names)
        torch.cuda.synchronize(self.device)      # the bind packs the filters with its own launches
        _lib.check(self._L.azx_train_bind(self._h, n, (C.c_char_p * n)(*names), (C.c_void_p * n)(*ptrs),
                                          (C.c_int64 * n)(*counts), (C.c_void_p * n)(*moms)))
        self._bound_key = self._key()

    # ---- stepping -----------------------------------------------------------------------------
    def _load(self, batch):
        k = batch["legal_moves"].shape[1]
        if len(batch["reward"]) != self.B:
            raise ValueError("NativeTrainStep was built for batches of %d rows, got %d" % (self.B, len(batch["reward"])))
        self.board.copy_(batch["board"].reshape(self.board.shape))
        self.reward.copy_(batch["reward"])
        self.legal_moves.zero_()
        self.moves_prob.zero_()
        self.legal_moves[:, :k].copy_(batch["legal_moves"])
        self.moves_prob[:, :k].copy_(batch["moves_prob"])

    def step(self, batch):
        """One training step on `batch` (the dict torch_batch_replays / DeviceReplayBuffer.sample produce); returns the
        three loss tensors (device; read them only when logging)."""
        self._load(batch)
        return self._run()

    def step_from_ring(self, replaybuf, indices):
        """One training step on the rows `indices` of a DeviceReplayBuffer: the collate kernel and the step are queued
        on torch's current stream, nothing waits for either (rows are full width: returns (loss, cells))."""
        replaybuf.collate_async(indices, dict(color=self._color, legal_moves=self.legal_moves, result=self._result,
                                              board=self.board.view(self.B, self.cells), moves_prob=self.moves_prob,
                                              reward=self.reward))
        return self._run(), self.cells

    def _quick_key(self):
        """Three addresses that move with any wholesale re-allocation (module.to(), optimizer.load_state_dict): checked
        every step; the full table (one data_ptr per tensor, ~0.1 ms of Python) every 64th."""
        p0, p1 = self._sentinels
        m0 = self.optimizer.state.get(p0, {}).get("momentum_buffer")
        return (p0.data_ptr(), p1.data_ptr(), None if m0 is None else m0.data_ptr())

    def rebind(self):
        """Call after re-allocating a single tensor of the module or the optimizer by hand (nothing in torch does)."""
        self._bound_key = None

    def _run(self):
        stale = self._bound_key is None or self._quick_key() != self._quick_bound
        if not stale and self.steps % 64 == 0:
            stale = self._key() != self._bound_key
        if stale:
            self._bind()          # first step, or the module / optimizer state was re-allocated (load_state_dict, .to())
            self._quick_bound = self._quick_key()
        g = self.optimizer.param_groups[0]
        self.model.train(True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.azx_train_step(self._h, float(g["lr"]), float(g.get("momentum", 0.0)),
                                          float(g.get("weight_decay", 0.0)), C.c_void_p(stream)))
        # the parameters moved without autograd noticing (Policy._sync_weights watches this counter)
        self.model.weight_updates_outside_autograd = getattr(self.model, "weight_updates_outside_autograd", 0) + 1
        self.steps += 1
        return self.loss

    def outputs(self, k: int):
        """(value, moves_logprob[:, :k]) of the last step, as Network.run returns them."""
        return {"value": self.out_value, "moves_logprob": self.out_logprob[:, :k]}

    def debug(self, name: str) -> np.ndarray:
        """An internal buffer by name (tests): raw<l> / act<l> / g<l> as float32 [B * cells * C], 'grad:<tensor>' flat."""
        nbytes = C.c_int64(0)
        _lib.check(self._L.azx_train_debug(self._h, name.encode(), None, 0, C.byref(nbytes)))
        dt = np.float64 if name in ("sums", "hsums") else np.float32
        out = np.empty(nbytes.value // np.dtype(dt).itemsize, dt)
        _lib.check(self._L.azx_train_debug(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), nbytes.value, C.byref(nbytes)))
        return out
