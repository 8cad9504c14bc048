"""Per-game POD <-> interventions-JSON codecs (slow path; the hot path never touches JSON)."""
from . import amidar, breakout, gridworld, space_invaders

CODECS = {"breakout": breakout, "space_invaders": space_invaders, "amidar": amidar, "gridworld": gridworld}


def codec(game_name):
    return CODECS[game_name]
