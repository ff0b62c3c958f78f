"""ctypes binding of libazx_hip.so (include/azx.h).

The HIP engine is the product: if the shared object is missing or no MI355X is visible the
calls fail loudly -- there is deliberately no CPU fallback (and nothing here imports oracle/).
"""
import ctypes as C
import logging
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libazx_hip.so")

MAX_BOARD = 13
CELL_STRIDE = 192
MAX_BATCH = 16
ROW_METRICS = 8          # AZX_ROW_METRICS


def record_bytes(cells):
    """AZX_RECORD_BYTES (include/azx.h): one fixed-size replay record."""
    return (16 + 5 * cells + 15) // 16 * 16

EVAL_RESNET, EVAL_UNIFORM, EVAL_UNIFORM_HASH, EVAL_EXTERNAL = 0, 1, 2, 3
FLAG_NO_COMPACT = 1


class AzxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("n_games", C.c_int32), ("simulations", C.c_int32),
                ("search_batch_size", C.c_int32), ("exploration_coef", C.c_float),
                ("exploration_depth", C.c_int32), ("noise_alpha", C.c_double),
                ("noise_scale", C.c_double), ("temperature", C.c_double),
                ("evaluator", C.c_int32), ("num_blocks", C.c_int32), ("base_chans", C.c_int32),
                ("nodes_per_game", C.c_int32), ("flags", C.c_int32), ("device", C.c_int32),
                ("seed", C.c_uint64), ("game_index_stride", C.c_int32), ("game_index_offset", C.c_int32)]


class PlayStats(C.Structure):
    _fields_ = [("positions", C.c_int64), ("games", C.c_int64), ("game_errors", C.c_int64),
                ("plies", C.c_int64), ("selects", C.c_int64), ("evals", C.c_int64),
                ("sum_depth", C.c_int64), ("sum_k_interior", C.c_int64),
                ("sum_k_leaf", C.c_int64), ("sum_search_value", C.c_double),
                ("sum_root_width", C.c_double), ("sum_action_logprob", C.c_double),
                ("sum_reward_last", C.c_double), ("seconds", C.c_double),
                ("mcts_seconds", C.c_double), ("mcts_launches", C.c_int64),
                ("net_seconds", C.c_double), ("net_launches", C.c_int64),
                ("mcts_kernel_launches", C.c_int64), ("sum_game_length", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# every symbol include/azx.h declares: (restype, argtypes)
SYMBOLS = {
    "azx_last_error": (C.c_char_p, []),
    "azx_version": (C.c_int, []),
    "azx_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "azx_destroy": (None, [_vp]),
    "azx_set_weights": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_vp), _i64p, C.c_int]),
    "azx_debug_weights": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, _i64p, C.c_char_p, C.c_int]),
    "azx_set_prior_table": (C.c_int, [_vp, _f32p, C.c_int]),
    "azx_reset": (C.c_int, [_vp, _i32p, C.c_int, _i32p, _i32p, C.c_int]),
    "azx_set_active": (C.c_int, [_vp, _i32p]),
    "azx_search": (C.c_int, [_vp, _f64p, C.c_int, C.c_int, C.c_double]),
    "azx_search_begin": (C.c_int, [_vp, _f64p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]),
    "azx_search_step": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "azx_get_leaves": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p, _i32p, C.POINTER(C.c_int)]),
    "azx_put_evals": (C.c_int, [_vp, C.c_int, _f32p, _f32p]),
    "azx_get_evals": (C.c_int, [_vp, C.c_int, _f32p, _f32p, C.POINTER(C.c_int)]),
    "azx_get_root": (C.c_int, [_vp, _i32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i32p, _f32p]),
    "azx_get_status": (C.c_int, [_vp, _i32p]),
    "azx_get_tree_nodes": (C.c_int, [_vp, _i32p]),
    "azx_get_games": (C.c_int, [_vp, _i32p, _i32p, _i32p, _i32p]),
    "azx_advance": (C.c_int, [_vp, _i32p]),
    "azx_tree_dump": (C.c_int, [_vp, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f32p, _f32p, _f32p, _i32p, _i32p]),
    "azx_forward": (C.c_int, [_vp, C.c_int, C.c_int, _i32p, _i32p, _f32p, _f32p]),
    "azx_hex_replay": (C.c_int, [C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, _i32p, _i32p, _u64p, _i32p]),
    "azx_play": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _i32p, _i32p, _i32p, _f32p, _f32p, _i64p, C.POINTER(PlayStats)]),
    "azx_play_row_metrics": (C.c_int, [_vp, C.c_int64, _f32p, _i64p]),
    "azx_play_steps": (C.c_int, [_vp, C.c_int64, C.POINTER(PlayStats)]),
    "azx_replay_create": (C.c_int, [_vp, C.c_int64]),
    "azx_replay_state": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "azx_replay_set_state": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "azx_replay_put": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, _i32p, _f32p, _f32p]),
    "azx_replay_fill": (C.c_int, [_vp, C.c_int64, C.c_int64, _i64p, C.POINTER(PlayStats)]),
    "azx_play_device": (C.c_int, [_vp, C.c_int64, C.c_int64, _i64p, C.POINTER(PlayStats)]),
    "azx_rows_pack": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    "azx_replay_put_records": (C.c_int, [_vp, C.c_int64, _vp]),
    "azx_replay_put_records_async": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "azx_reserve_cus": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "azx_replay_collate": (C.c_int, [_vp, C.c_int64, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _i32p]),
    "azx_replay_collate_async": (C.c_int, [_vp, C.c_int64, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "azx_replay_set_mover_view": (C.c_int, [_vp, C.c_int]),
    "azx_selftest_arith": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p]),
    "azx_selftest_divide": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p]),
    "azx_selftest_dirichlet": (C.c_int, [C.c_int, C.c_double, C.c_int, C.c_int, C.c_uint32, _f32p]),
    "azx_debug_choose": (C.c_int, [_vp, _i32p, _f32p]),
    "azx_debug_counters": (C.c_int, [_vp, _u64p]),
    "azx_debug_counters_raw": (C.c_int, [_vp, _u64p, C.c_int64]),
    "azx_kernel_info": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "azx_debug_set_queue_cap": (C.c_int, [_vp, C.c_int64]),
    "azx_stream": (_vp, [_vp]),
}

class TrainConfig(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("num_blocks", C.c_int32), ("base_chans", C.c_int32),
                ("batch_size", C.c_int32), ("device", C.c_int32)]


_pp = C.POINTER(_vp)
SYMBOLS.update({
    "azx_train_create": (C.c_int, [C.POINTER(TrainConfig), C.POINTER(_vp)]),
    "azx_train_destroy": (None, [_vp]),
    "azx_train_bind": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), _pp, _i64p, _pp]),
    "azx_train_inputs": (C.c_int, [_vp, _pp, _pp, _pp, _pp]),
    "azx_train_outputs": (C.c_int, [_vp, _pp, _pp, _pp]),
    "azx_train_step": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, _vp]),
    "azx_train_debug": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64, _i64p]),
})

_lib = None


def build(force=False):
    """Compile libazx_hip.so for gfx950 with hipcc (azalea_amd/csrc/Makefile)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load the HIP engine; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AzxError(
                "azalea_amd/libazx_hip.so is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C azalea_amd/csrc`. "
                "The engine is HIP-only; there is no CPU fallback.")
        # PyTorch-ROCm ships its own HIP / HSA runtime.  If this library (linked against /opt/rocm's) brings the GPU up
        # first, a later first use of torch.cuda in the same process fails with "No HIP GPUs are available" -- and
        # every Policy holds its weights in torch tensors.  So where torch is installed and sees a GPU, its runtime is
        # initialised before ours is loaded; the order then no longer depends on what the caller touched first.
        # (AZX_NO_TORCH_PREINIT=1 skips this for consumers of the bare C API that never touch torch.)
        if not os.environ.get("AZX_NO_TORCH_PREINIT"):
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.init()
            except (ImportError, RuntimeError) as exc:      # no torch / no usable GPU: the engine reports AZX_ENODEV where it matters
                logging.getLogger(__name__).debug("torch pre-initialisation skipped: %r", exc)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise AzxError("azx error %d: %s" % (rc, lib().azx_last_error().decode(errors="replace")))
