from .hex import HexGame, HexGameState  # noqa: F401
