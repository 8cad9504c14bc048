"""Per-game POD <-> interventions-JSON codecs (slow path; the hot path never touches JSON)."""
from . import amidar, breakout, space_invaders

CODECS = {"breakout": breakout, "space_invaders": space_invaders, "amidar": amidar}


def codec(game_name):
    return CODECS[game_name]
