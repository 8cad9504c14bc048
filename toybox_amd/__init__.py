"""toybox_amd -- MI355X-native batched engine for Toybox's game-step hot path.

Host side of the C-ABI in include/toybox_amd.h (hand-written HIP for gfx950 in toybox_amd/csrc).
"""
from ._abi import GAME_IDS, GAME_NAMES  # noqa: F401
from ._lib import ToyboxAmdError  # noqa: F401
from .engine import Engine  # noqa: F401

__version__ = "0.1.0"
