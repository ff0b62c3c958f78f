"""azalea_amd -- MI355X-native batched Hex self-play engine behind azalea's Policy / AzaleaAgent /
Player / ReplayDataFrame surface.  The search, rules and network forward run as hand-written
gfx950 HIP kernels in libazx_hip.so (see include/azx.h); this package is the thin host side."""
from .version import __version__  # noqa: F401

_LAZY = {
    "Engine": ("engine", "Engine"),
    "AzaleaAgent": ("azalea_agent", "AzaleaAgent"),
    "Policy": ("policy", "Policy"),
    "RandomPolicy": ("random_policy", "RandomPolicy"),
    "Player": ("parallel_player", "Player"),
    "ReplayBuffer": ("replay_buffer", "ReplayBuffer"),
    "ReplayDataFrame": ("replay_buffer", "ReplayDataFrame"),
    "ReplayRecord": ("replay_buffer", "ReplayRecord"),
    "DeviceReplayBuffer": ("device_replay", "DeviceReplayBuffer"),
    "HexGame": ("game.hex", "HexGame"),
    "HexNetwork": ("network", "HexNetwork"),
    "SearchTreeFull": ("policy", "SearchTreeFull"),
    "compute_ranking": ("ranking", "compute_ranking"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
