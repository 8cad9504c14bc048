"""Breakout: POD records <-> the interventions JSON schema.

State key set (ctoybox 0.5.0 era) == the kwargs of /root/reference/toybox/interventions/breakout.py:49-54
and Brick.expected_keys (:198), Ball (:276), Paddle (:132); config keys == the golden dump
toybox/interventions/defaults/breakout_config_default.json.  Missing keys raise; unknown keys are
ignored (serde's default on the Rust side -- the strict check lives in the reference's own
BaseMixin.decode, interventions/base.py:209-225, which runs before the engine sees the JSON).
"""
import math

from .. import _abi
from .._abi import BreakoutConfig, BreakoutState, Color

STATE_KEYS = ["score", "lives", "rand", "level", "paddle", "paddle_width", "paddle_speed", "ball_radius",
              "balls", "bricks", "reset", "is_dead"]
BRICK_KEYS = ["destructible", "depth", "color", "alive", "points", "size", "position", "row", "col"]
CONFIG_KEYS = ["paddle_discrete_segments", "ball_start_positions", "start_lives", "row_scores",
               "ball_speed_row_depth", "bg_color", "rand", "row_colors", "frame_color", "paddle_color",
               "ball_color", "ball_speed_fast", "ball_speed_slow"]


def _strict(d, keys, what):
    """Required keys must be present; unknown keys are ignored, as serde does on the Rust side (the reference's own
    MovementAI.encode leaks `_in_init` / `schema` into the AI parameters, interventions/amidar.py:159-164)."""
    missing = set(keys) - set(d.keys())
    if missing:
        raise ValueError("%s: missing keys %s" % (what, sorted(missing)))


def _vec(x, y):
    return {"x": float(x), "y": float(y)}


# ------------------------------------------------------------------ config

def fill_trig(cfg):
    """Host-evaluated trig tables (libm via math.cos/sin) -- the device never evaluates trig."""
    for i in range(cfg.n_starts):
        rad = cfg.start_angle_deg[i] * (math.pi / 180.0)
        cfg.start_dir_x[i] = math.cos(rad)
        cfg.start_dir_y[i] = math.sin(rad)
    s = cfg.paddle_discrete_segments
    for i in range(max(0, min(s, _abi.BRK_MAX_SEGMENTS))):
        deg = 90.0 if s == 1 else 150.0 - float(i) * (120.0 / float(s - 1))
        rad = deg * (math.pi / 180.0)
        cfg.paddle_dir_x[i] = math.cos(rad)
        cfg.paddle_dir_y[i] = -math.sin(rad)


def config_from_json(d):
    _strict(d, CONFIG_KEYS, "breakout config")
    cfg = BreakoutConfig()
    cfg.rand[0], cfg.rand[1] = (int(v) for v in d["rand"]["state"])
    cfg.start_lives = int(d["start_lives"])
    scores, colors = d["row_scores"], d["row_colors"]
    if len(scores) != len(colors):
        raise ValueError("breakout config: row_scores and row_colors differ in length")
    if not 1 <= len(scores) <= _abi.BRK_MAX_ROWS:
        raise ValueError("breakout config: 1..%d brick rows supported" % _abi.BRK_MAX_ROWS)
    cfg.n_rows = len(scores)
    for i, (s, c) in enumerate(zip(scores, colors)):
        cfg.row_scores[i] = int(s)
        cfg.row_colors[i] = Color.from_json(c)
    cfg.ball_speed_row_depth = int(d["ball_speed_row_depth"])
    cfg.ball_speed_slow = float(d["ball_speed_slow"])
    cfg.ball_speed_fast = float(d["ball_speed_fast"])
    starts = d["ball_start_positions"]
    if not 1 <= len(starts) <= _abi.BRK_MAX_STARTS:
        raise ValueError("breakout config: 1..%d ball_start_positions supported" % _abi.BRK_MAX_STARTS)
    cfg.n_starts = len(starts)
    for i, s in enumerate(starts):
        _strict(s, ["x", "y", "angle_degrees"], "ball_start_position")
        cfg.start_x[i], cfg.start_y[i], cfg.start_angle_deg[i] = float(s["x"]), float(s["y"]), float(s["angle_degrees"])
    seg = int(d["paddle_discrete_segments"])
    if not 1 <= seg <= _abi.BRK_MAX_SEGMENTS:
        raise ValueError("breakout config: paddle_discrete_segments must be 1..%d on the device engine" % _abi.BRK_MAX_SEGMENTS)
    cfg.paddle_discrete_segments = seg
    for k in ("bg_color", "frame_color", "paddle_color", "ball_color"):
        setattr(cfg, k, Color.from_json(d[k]))
    fill_trig(cfg)
    return cfg


def config_to_json(cfg):
    return {
        "paddle_discrete_segments": cfg.paddle_discrete_segments,
        "ball_start_positions": [{"angle_degrees": cfg.start_angle_deg[i], "y": cfg.start_y[i], "x": cfg.start_x[i]}
                                 for i in range(cfg.n_starts)],
        "start_lives": cfg.start_lives,
        "row_scores": [cfg.row_scores[i] for i in range(cfg.n_rows)],
        "ball_speed_row_depth": cfg.ball_speed_row_depth,
        "bg_color": cfg.bg_color.to_json(),
        "rand": {"state": [int(cfg.rand[0]), int(cfg.rand[1])]},
        "row_colors": [cfg.row_colors[i].to_json() for i in range(cfg.n_rows)],
        "frame_color": cfg.frame_color.to_json(),
        "paddle_color": cfg.paddle_color.to_json(),
        "ball_color": cfg.ball_color.to_json(),
        "ball_speed_fast": cfg.ball_speed_fast,
        "ball_speed_slow": cfg.ball_speed_slow,
    }


def default_config():
    """== toybox/interventions/defaults/breakout_config_default.json with rand = seed(13)."""
    rgb = lambda r, g, b: {"r": r, "g": g, "b": b, "a": 255}
    return config_from_json({
        "paddle_discrete_segments": 5,
        "ball_start_positions": [{"angle_degrees": 30.0, "y": 80.0, "x": 24.0}, {"angle_degrees": 30.0, "y": 80.0, "x": 120.0},
                                 {"angle_degrees": 150.0, "y": 80.0, "x": 120.0}, {"angle_degrees": 150.0, "y": 80.0, "x": 216.0}],
        "start_lives": 5,
        "row_scores": [7, 7, 4, 4, 1, 1],
        "ball_speed_row_depth": 3,
        "bg_color": rgb(0, 0, 0),
        "rand": {"state": [0x193A6754A8A7D469 ^ 13, 0x97830E05113BA7BB]},
        "row_colors": [rgb(200, 72, 72), rgb(198, 108, 58), rgb(180, 122, 48), rgb(162, 162, 42), rgb(72, 160, 72), rgb(66, 72, 200)],
        "frame_color": rgb(144, 144, 144),
        "paddle_color": rgb(200, 72, 72),
        "ball_color": rgb(200, 72, 72),
        "ball_speed_fast": 4.0,
        "ball_speed_slow": 2.0,
    })


# ------------------------------------------------------------------ state

def state_to_json(st):
    bricks = []
    for i in range(st.n_bricks):
        b = st.bricks[i]
        bricks.append({
            "destructible": bool(b.destructible), "depth": b.depth, "color": b.color.to_json(),
            "alive": bool(b.alive), "points": b.points, "size": _vec(b.w, b.h), "position": _vec(b.x, b.y),
            "row": b.row, "col": b.col,
        })
    return {
        "score": st.score, "lives": st.lives, "rand": {"state": [int(st.rand[0]), int(st.rand[1])]}, "level": st.level,
        "paddle": {"velocity": _vec(st.paddle_vx, st.paddle_vy), "position": _vec(st.paddle_x, st.paddle_y)},
        "paddle_width": st.paddle_width, "paddle_speed": st.paddle_speed, "ball_radius": st.ball_radius,
        "balls": [{"position": _vec(st.ball_x[i], st.ball_y[i]), "velocity": _vec(st.ball_vx[i], st.ball_vy[i])}
                  for i in range(st.n_balls)],
        "bricks": bricks,
        "reset": bool(st.reset), "is_dead": bool(st.is_dead),
    }


def state_from_json(d):
    _strict(d, STATE_KEYS, "breakout state")
    st = BreakoutState()
    st.rand[0], st.rand[1] = (int(v) for v in d["rand"]["state"])
    st.score, st.lives, st.level = int(d["score"]), int(d["lives"]), int(d["level"])
    st.is_dead, st.reset = int(bool(d["is_dead"])), int(bool(d["reset"]))
    _strict(d["paddle"], ["velocity", "position"], "paddle")
    st.paddle_x, st.paddle_y = float(d["paddle"]["position"]["x"]), float(d["paddle"]["position"]["y"])
    st.paddle_vx, st.paddle_vy = float(d["paddle"]["velocity"]["x"]), float(d["paddle"]["velocity"]["y"])
    st.paddle_width, st.paddle_speed, st.ball_radius = float(d["paddle_width"]), float(d["paddle_speed"]), float(d["ball_radius"])
    balls = d["balls"]
    if len(balls) > _abi.BRK_MAX_BALLS:
        raise ValueError("breakout state: at most %d balls on the device engine" % _abi.BRK_MAX_BALLS)
    st.n_balls = len(balls)
    for i, b in enumerate(balls):
        _strict(b, ["position", "velocity"], "ball")
        st.ball_x[i], st.ball_y[i] = float(b["position"]["x"]), float(b["position"]["y"])
        st.ball_vx[i], st.ball_vy[i] = float(b["velocity"]["x"]), float(b["velocity"]["y"])
    bricks = d["bricks"]
    if len(bricks) > _abi.BRK_MAX_BRICKS:
        raise ValueError("breakout state: at most %d bricks on the device engine" % _abi.BRK_MAX_BRICKS)
    st.n_bricks = len(bricks)
    for i, b in enumerate(bricks):
        _strict(b, BRICK_KEYS, "brick")
        k = st.bricks[i]
        k.x, k.y = float(b["position"]["x"]), float(b["position"]["y"])
        k.w, k.h = float(b["size"]["x"]), float(b["size"]["y"])
        k.points, k.depth, k.row, k.col = int(b["points"]), int(b["depth"]), int(b["row"]), int(b["col"])
        k.color = Color.from_json(b["color"])
        k.alive, k.destructible = int(bool(b["alive"])), int(bool(b["destructible"]))
    return st


def schema_for_state():
    num = {"type": "number", "format": "double"}
    integer = {"type": "integer", "format": "int32"}
    boolean = {"type": "boolean"}
    vec = {"type": "object", "required": ["x", "y"], "properties": {"x": num, "y": num}}
    color = {"type": "object", "required": ["r", "g", "b", "a"],
             "properties": {k: {"type": "integer", "format": "uint8"} for k in "rgba"}}
    body = {"type": "object", "required": ["position", "velocity"], "properties": {"position": vec, "velocity": vec}}
    brick = {"type": "object", "required": list(BRICK_KEYS), "properties": {
        "destructible": boolean, "depth": integer, "color": color, "alive": boolean, "points": integer,
        "size": vec, "position": vec, "row": integer, "col": integer}}
    props = {
        "score": integer, "lives": integer, "level": integer,
        "rand": {"type": "object", "required": ["state"],
                 "properties": {"state": {"type": "array", "items": {"type": "integer", "format": "uint64"}}}},
        "paddle": body, "paddle_width": num, "paddle_speed": num, "ball_radius": num,
        "balls": {"type": "array", "items": body}, "bricks": {"type": "array", "items": brick},
        "reset": boolean, "is_dead": boolean,
    }
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "Breakout", "type": "object",
            "required": list(STATE_KEYS), "properties": props}


def schema_for_config():
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "BreakoutConfig", "type": "object",
            "required": list(CONFIG_KEYS), "properties": {}}


def query(tb, name, args):
    """State queries of the Breakout core (slow path, evaluated from the state JSON)."""
    js = tb.state_to_json()
    bricks = js["bricks"]
    rows = max((b["row"] for b in bricks), default=-1) + 1
    cols = max((b["col"] for b in bricks), default=-1) + 1
    if name == "bricks_remaining":
        return sum(1 for b in bricks if b["alive"])
    if name in ("count_channels", "channel_count"):
        return sum(1 for c in range(cols) if all(not b["alive"] for b in bricks if b["col"] == c))
    if name == "num_columns":
        return cols
    if name == "num_rows":
        return rows
    if name == "channels":
        return [c for c in range(cols) if all(not b["alive"] for b in bricks if b["col"] == c)]
    if name == "brick_live_by_index":
        return bool(bricks[int(args)]["alive"])
    raise ValueError("unknown breakout query %r" % (name,))
