"""Per-game POD <-> interventions-JSON codecs (slow path; the hot path never touches JSON)."""
from . import breakout

CODECS = {"breakout": breakout}


def codec(game_name):
    return CODECS[game_name]
