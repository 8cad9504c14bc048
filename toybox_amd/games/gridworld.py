"""GridWorld: POD records <-> the JSON of the reference's two dumps.

Key sets == toybox/interventions/defaults/gridworld_config_default.json (config: tiles keyed by one-character names,
grid rows as strings of those names) and gridworld_state_default.json (state: tiles as a list, grid as rows of indices
into it).  The reference has no intervention class for this game (envs/atari/gridworld.py:8-13 is its only user).
"""
from .. import _abi
from .._abi import Color, GridWorldConfig, GridWorldState

CONFIG_KEYS = ["reward_becomes", "grid", "player_start", "player_color", "game_size", "tiles"]
STATE_KEYS = ["reward_becomes", "grid", "score", "player_color", "game_over", "player", "tiles"]
TILE_KEYS = ["color", "goal", "reward", "walkable"]
D = _abi.GW_MAX_DIM


def _strict(d, keys, what):
    missing = set(keys) - set(d.keys())
    if missing:
        raise ValueError("%s: missing keys %s" % (what, sorted(missing)))


def _tile_from_json(t, rec):
    _strict(t, TILE_KEYS, "gridworld tile")
    rec.color = Color.from_json(t["color"])
    rec.reward = int(t["reward"])
    rec.goal, rec.walkable = int(bool(t["goal"])), int(bool(t["walkable"]))


def _tile_to_json(rec):
    return {"color": rec.color.to_json(), "goal": bool(rec.goal), "reward": int(rec.reward), "walkable": bool(rec.walkable)}


def _check_size(w, h, what):
    if not (1 <= w <= D and 1 <= h <= D):
        raise ValueError("%s: game_size must be 1..%d x 1..%d, not %dx%d" % (what, D, D, w, h))


def config_from_json(d):
    _strict(d, CONFIG_KEYS, "gridworld config")
    cfg = GridWorldConfig()
    if "rand" in d:
        cfg.rand[0], cfg.rand[1] = (int(v) for v in d["rand"]["state"])
    w, h = (int(v) for v in d["game_size"])
    _check_size(w, h, "gridworld config")
    cfg.width, cfg.height = w, h
    names = list(d["tiles"].keys())
    if not 1 <= len(names) <= _abi.GW_MAX_TILES:
        raise ValueError("gridworld config: 1..%d tiles supported" % _abi.GW_MAX_TILES)
    for i, name in enumerate(names):
        if len(name) != 1 or ord(name) > 255:
            raise ValueError("gridworld config: tile names are single characters, not %r" % (name,))
        cfg.tile_keys[i] = ord(name)
        _tile_from_json(d["tiles"][name], cfg.tiles[i])
    cfg.n_tiles = len(names)
    index = {name: i for i, name in enumerate(names)}
    if str(d["reward_becomes"]) not in index:
        raise ValueError("gridworld config: reward_becomes names no tile")
    cfg.reward_becomes = index[str(d["reward_becomes"])]
    rows = d["grid"]
    if len(rows) != h or any(len(r) != w for r in rows):
        raise ValueError("gridworld config: grid does not match game_size")
    for y, row in enumerate(rows):
        for x, ch in enumerate(row):
            if ch not in index:
                raise ValueError("gridworld config: grid uses unknown tile %r" % (ch,))
            cfg.grid[y * D + x] = index[ch]
    cfg.player_start_x, cfg.player_start_y = (int(v) for v in d["player_start"])
    cfg.player_color = Color.from_json(d["player_color"])
    return cfg


def config_to_json(cfg):
    names = [chr(cfg.tile_keys[i]) for i in range(cfg.n_tiles)]
    return {
        "reward_becomes": names[cfg.reward_becomes],
        "grid": ["".join(names[cfg.grid[y * D + x]] for x in range(cfg.width)) for y in range(cfg.height)],
        "player_start": [int(cfg.player_start_x), int(cfg.player_start_y)],
        "player_color": cfg.player_color.to_json(),
        "game_size": [int(cfg.width), int(cfg.height)],
        "tiles": {names[i]: _tile_to_json(cfg.tiles[i]) for i in range(cfg.n_tiles)},
    }


def default_config():
    return {
        "reward_becomes": "0",
        "grid": ["111111111", "1000R0001", "101111101", "100010001", "10001R111", "1000100G1", "111111111"],
        "player_start": [2, 4],
        "player_color": {"r": 255, "g": 0, "a": 255, "b": 0},
        "game_size": [9, 7],
        "tiles": {
            "0": {"color": {"r": 255, "g": 255, "a": 255, "b": 255}, "goal": False, "reward": 0, "walkable": True},
            "1": {"color": {"r": 0, "g": 0, "a": 255, "b": 0}, "goal": False, "reward": 0, "walkable": False},
            "G": {"color": {"r": 0, "g": 255, "a": 255, "b": 0}, "goal": True, "reward": 10, "walkable": True},
            "R": {"color": {"r": 255, "g": 255, "a": 255, "b": 0}, "goal": False, "reward": 1, "walkable": True},
        },
    }


def state_to_json(st):
    return {
        "reward_becomes": int(st.reward_becomes),
        "grid": [[int(st.grid[y * D + x]) for x in range(st.width)] for y in range(st.height)],
        "score": int(st.score),
        "player_color": st.player_color.to_json(),
        "game_over": bool(st.game_over),
        "player": [int(st.player_x), int(st.player_y)],
        "tiles": [_tile_to_json(st.tiles[i]) for i in range(st.n_tiles)],
    }


def state_from_json(d):
    _strict(d, STATE_KEYS, "gridworld state")
    st = GridWorldState()
    rows = d["grid"]
    h = len(rows)
    w = len(rows[0]) if h else 0
    _check_size(w, h, "gridworld state")
    if any(len(r) != w for r in rows):
        raise ValueError("gridworld state: ragged grid")
    tiles = d["tiles"]
    if not 1 <= len(tiles) <= _abi.GW_MAX_TILES:
        raise ValueError("gridworld state: 1..%d tiles supported" % _abi.GW_MAX_TILES)
    st.width, st.height, st.n_tiles = w, h, len(tiles)
    for i, t in enumerate(tiles):
        _tile_from_json(t, st.tiles[i])
    for y, row in enumerate(rows):
        for x, v in enumerate(row):
            if not 0 <= int(v) < len(tiles):
                raise ValueError("gridworld state: grid cell (%d,%d) names no tile" % (x, y))
            st.grid[y * D + x] = int(v)
    if not 0 <= int(d["reward_becomes"]) < len(tiles):
        raise ValueError("gridworld state: reward_becomes names no tile")
    st.reward_becomes = int(d["reward_becomes"])
    st.score = int(d["score"])
    st.game_over = int(bool(d["game_over"]))
    st.player_x, st.player_y = (int(v) for v in d["player"])
    st.player_color = Color.from_json(d["player_color"])
    return st


def schema_for_state():
    integer = {"type": "integer", "format": "int32"}
    color = {"type": "object", "required": ["r", "g", "b", "a"],
             "properties": {k: {"type": "integer", "format": "uint8"} for k in "rgba"}}
    tile = {"type": "object", "required": list(TILE_KEYS),
            "properties": {"color": color, "goal": {"type": "boolean"}, "reward": integer, "walkable": {"type": "boolean"}}}
    props = {"reward_becomes": integer, "score": integer, "game_over": {"type": "boolean"}, "player_color": color,
             "player": {"type": "array", "items": integer}, "tiles": {"type": "array", "items": tile},
             "grid": {"type": "array", "items": {"type": "array", "items": integer}}}
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "GridWorld", "type": "object",
            "required": list(STATE_KEYS), "properties": props}


def schema_for_config():
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "GridWorldConfig", "type": "object",
            "required": list(CONFIG_KEYS), "properties": {}}


def query(tb, name, args):
    js = tb.state_to_json()
    if name == "xy":
        return list(js["player"])
    if name == "xyt":
        x, y = js["player"]
        return [x, y, js["grid"][y][x]]
    raise ValueError("gridworld: unknown query %r" % (name,))
