"""Amidar: POD records <-> the interventions JSON schema.

State key set == the kwargs of /root/reference/toybox/interventions/amidar.py:22-24 with Enemy/Player (:171,:195),
MovementAI protocols and their parameter sets (:101-112, :421-448), Board (:216), Box (:300), TilePoint, WorldPoint;
config keys == the golden dump toybox/interventions/defaults/amidar_config_default.json.  `junctions` is derived
from the tiles (walkable tiles with both a horizontal and a vertical walkable neighbour) and, like the golden's,
carries no order: compare it as a set.  None-valued AI parameters may be absent on input
(MovementAI.encode omits them, interventions/amidar.py:159-164).
"""
from .. import _abi
from .._abi import AmidarAI, AmidarConfig, AmidarState, Color

STATE_KEYS = ["score", "player", "lives", "rand", "level", "enemies", "jumps", "jump_timer", "chase_timer", "board"]
MOVER_KEYS = ["history", "step", "position", "caught", "speed", "ai"]
BOARD_KEYS = ["boxes", "tiles", "height", "chase_junctions", "width", "junctions"]
BOX_KEYS = ["triggers_chase", "top_left", "bottom_right", "painted"]
CONFIG_KEYS = ["box_bonus", "inner_painted_color", "jump_time", "render_images", "board", "enemy_color", "chase_time",
               "rand", "painted_color", "enemies", "start_lives", "player_start", "start_jumps", "default_board_bugs",
               "player_color", "bg_color", "chase_score_bonus", "unpainted_color"]
COLOR_KEYS = ["bg_color", "player_color", "unpainted_color", "painted_color", "enemy_color", "inner_painted_color"]
BOARD_CHARS = {" ": 0, "=": 1, "p": 2, "c": 3}
BOARD_CHARS_INV = {v: k for k, v in BOARD_CHARS.items()}
AI_PARAMS = {
    "EnemyLookupAI": ["next", "default_route_index"],
    "EnemyPerimeterAI": ["start"],
    "EnemyAmidarMvmt": ["vert", "horiz", "start_vert", "start_horiz", "start"],
    "EnemyTargetPlayer": ["start", "start_dir", "vision_distance", "dir", "player_seen"],
    "EnemyRandomMvmt": ["start", "start_dir", "dir"],
}
OPTIONAL_AI_PARAMS = {"player_seen"}


def _strict(d, keys, what):
    """Required keys must be present; unknown keys are ignored, as serde does on the Rust side (the reference's own
    MovementAI.encode leaks `_in_init` / `schema` into the AI parameters, interventions/amidar.py:159-164)."""
    missing = set(keys) - set(d.keys())
    if missing:
        raise ValueError("%s: missing keys %s" % (what, sorted(missing)))


def _dir(v):
    return _abi.DIR_NAMES.index(v)


# ------------------------------------------------------------------ AI

def ai_from_json(d):
    ai = AmidarAI()
    ai.seen_tx = ai.seen_ty = -1
    if d == "Player":
        ai.kind = 0
        return ai
    if not isinstance(d, dict) or len(d) != 1:
        raise ValueError("amidar: ai must be 'Player' or {<protocol>: {...}}")
    name, p = next(iter(d.items()))
    if name not in AI_PARAMS:
        raise ValueError("amidar: unknown movement protocol %r" % (name,))
    missing = set(AI_PARAMS[name]) - set(p) - OPTIONAL_AI_PARAMS
    if missing:
        raise ValueError("amidar %s: missing parameters %s" % (name, sorted(missing)))
    ai.kind = _abi.AI_NAMES.index(name)
    if "next" in p:
        ai.next, ai.default_route_index = int(p["next"]), int(p["default_route_index"])
    if "start" in p:
        ai.start_tx, ai.start_ty = int(p["start"]["tx"]), int(p["start"]["ty"])
    for k in ("vert", "horiz", "start_vert", "start_horiz", "start_dir", "dir"):
        if k in p:
            setattr(ai, k, _dir(p[k]))
    if "vision_distance" in p:
        ai.vision_distance = int(p["vision_distance"])
    if p.get("player_seen") is not None:
        ai.seen_tx, ai.seen_ty = int(p["player_seen"]["tx"]), int(p["player_seen"]["ty"])
    return ai


def ai_to_json(ai):
    name = _abi.AI_NAMES[ai.kind]
    if name == "Player":
        return "Player"
    tp = lambda x, y: {"tx": x, "ty": y}
    dn = lambda v: _abi.DIR_NAMES[v & 3]
    full = {
        "next": ai.next, "default_route_index": ai.default_route_index, "start": tp(ai.start_tx, ai.start_ty),
        "vert": dn(ai.vert), "horiz": dn(ai.horiz), "start_vert": dn(ai.start_vert), "start_horiz": dn(ai.start_horiz),
        "start_dir": dn(ai.start_dir), "dir": dn(ai.dir), "vision_distance": ai.vision_distance,
        "player_seen": tp(ai.seen_tx, ai.seen_ty) if ai.seen_tx >= 0 else None,
    }
    return {name: {k: full[k] for k in AI_PARAMS[name]}}


# ------------------------------------------------------------------ config

def config_from_json(d):
    _strict(d, CONFIG_KEYS, "amidar config")
    cfg = AmidarConfig()
    cfg.rand[0], cfg.rand[1] = (int(v) for v in d["rand"]["state"])
    for k in ("start_lives", "start_jumps", "jump_time", "chase_time", "box_bonus", "chase_score_bonus"):
        setattr(cfg, k, int(d[k]))
    cfg.player_start_tx, cfg.player_start_ty = int(d["player_start"]["tx"]), int(d["player_start"]["ty"])
    cfg.render_images, cfg.default_board_bugs = int(bool(d["render_images"])), int(bool(d["default_board_bugs"]))
    enemies = d["enemies"]
    if len(enemies) > _abi.AMI_MAX_ENEMIES:
        raise ValueError("amidar config: at most %d enemies on the device engine" % _abi.AMI_MAX_ENEMIES)
    cfg.n_enemies = len(enemies)
    for i, e in enumerate(enemies):
        cfg.enemies[i] = ai_from_json(e)
    for k in COLOR_KEYS:
        setattr(cfg, k, Color.from_json(d[k]))
    board = d["board"]
    if len(board) != _abi.AMI_BOARD_H or any(len(r) != _abi.AMI_BOARD_W for r in board):
        raise ValueError("amidar config: the board is %d rows of %d tiles" % (_abi.AMI_BOARD_H, _abi.AMI_BOARD_W))
    for y, row in enumerate(board):
        for x, ch in enumerate(row):
            if ch not in BOARD_CHARS:
                raise ValueError("amidar config: unknown board character %r" % ch)
            cfg.board[y][x] = BOARD_CHARS[ch]
    return cfg


def config_to_json(cfg):
    out = {k: getattr(cfg, k) for k in ("start_lives", "start_jumps", "jump_time", "chase_time", "box_bonus", "chase_score_bonus")}
    out["player_start"] = {"tx": cfg.player_start_tx, "ty": cfg.player_start_ty}
    out["render_images"], out["default_board_bugs"] = bool(cfg.render_images), bool(cfg.default_board_bugs)
    out["enemies"] = [ai_to_json(cfg.enemies[i]) for i in range(cfg.n_enemies)]
    for k in COLOR_KEYS:
        out[k] = getattr(cfg, k).to_json()
    out["board"] = ["".join(BOARD_CHARS_INV[cfg.board[y][x]] for x in range(_abi.AMI_BOARD_W)) for y in range(_abi.AMI_BOARD_H)]
    out["rand"] = {"state": [int(cfg.rand[0]), int(cfg.rand[1])]}
    return out


# ------------------------------------------------------------------ state

def _mover_to_json(m):
    return {"history": [m.history[i] for i in range(m.n_history)],
            "step": {"tx": m.step_tx, "ty": m.step_ty} if m.step_tx >= 0 else None,
            "position": {"x": m.x, "y": m.y}, "caught": bool(m.caught), "speed": m.speed, "ai": ai_to_json(m.ai)}


def _mover_from_json(d, m):
    _strict(d, MOVER_KEYS, "amidar mover")
    hist = list(d["history"])
    if len(hist) > _abi.AMI_MAX_HISTORY:
        hist = hist[-_abi.AMI_MAX_HISTORY:]      # the engine keeps the most recent junctions
    m.n_history = len(hist)
    for i, h in enumerate(hist):
        m.history[i] = int(h)
    if d["step"] is None:
        m.step_tx = m.step_ty = -1
    else:
        m.step_tx, m.step_ty = int(d["step"]["tx"]), int(d["step"]["ty"])
    m.x, m.y = int(d["position"]["x"]), int(d["position"]["y"])
    m.caught, m.speed = int(bool(d["caught"])), int(d["speed"])
    m.ai = ai_from_json(d["ai"])


def junctions_of(tiles):
    H, W = _abi.AMI_BOARD_H, _abi.AMI_BOARD_W
    walk = lambda x, y: 0 <= x < W and 0 <= y < H and tiles[y][x] != 0
    return [y * W + x for y in range(H) for x in range(W)
            if walk(x, y) and (walk(x - 1, y) or walk(x + 1, y)) and (walk(x, y - 1) or walk(x, y + 1))]


def state_to_json(st):
    tiles = [[st.tiles[y][x] for x in range(_abi.AMI_BOARD_W)] for y in range(_abi.AMI_BOARD_H)]
    return {
        "score": st.score, "lives": st.lives, "level": st.level, "rand": {"state": [int(st.rand[0]), int(st.rand[1])]},
        "jumps": st.jumps, "jump_timer": st.jump_timer, "chase_timer": st.chase_timer,
        "player": _mover_to_json(st.player),
        "enemies": [_mover_to_json(st.enemies[i]) for i in range(st.n_enemies)],
        "board": {
            "boxes": [{"triggers_chase": bool(b.triggers_chase), "top_left": {"tx": b.tl_tx, "ty": b.tl_ty},
                       "bottom_right": {"tx": b.br_tx, "ty": b.br_ty}, "painted": bool(b.painted)}
                      for b in (st.boxes[i] for i in range(st.n_boxes))],
            "tiles": [[_abi.TILE_NAMES[t] for t in row] for row in tiles],
            "height": _abi.AMI_BOARD_H, "width": _abi.AMI_BOARD_W,
            "chase_junctions": [st.chase_junctions[i] for i in range(st.n_chase_junctions)],
            "junctions": junctions_of(tiles),
        },
    }


def state_from_json(d):
    _strict(d, STATE_KEYS, "amidar state")
    st = AmidarState()
    st.rand[0], st.rand[1] = (int(v) for v in d["rand"]["state"])
    for k in ("score", "lives", "level", "jumps", "jump_timer", "chase_timer"):
        setattr(st, k, int(d[k]))
    _mover_from_json(d["player"], st.player)
    enemies = d["enemies"]
    if len(enemies) > _abi.AMI_MAX_ENEMIES:
        raise ValueError("amidar state: at most %d enemies on the device engine" % _abi.AMI_MAX_ENEMIES)
    st.n_enemies = len(enemies)
    for i, e in enumerate(enemies):
        _mover_from_json(e, st.enemies[i])
    b = d["board"]
    _strict(b, BOARD_KEYS, "amidar board")
    if int(b["width"]) != _abi.AMI_BOARD_W or int(b["height"]) != _abi.AMI_BOARD_H:
        raise ValueError("amidar state: the board is %dx%d tiles" % (_abi.AMI_BOARD_W, _abi.AMI_BOARD_H))
    tiles = b["tiles"]
    if len(tiles) != _abi.AMI_BOARD_H or any(len(r) != _abi.AMI_BOARD_W for r in tiles):
        raise ValueError("amidar state: tiles must be %d rows of %d" % (_abi.AMI_BOARD_H, _abi.AMI_BOARD_W))
    for y, row in enumerate(tiles):
        for x, t in enumerate(row):
            st.tiles[y][x] = _abi.TILE_NAMES.index(t)
    boxes = b["boxes"]
    if len(boxes) > _abi.AMI_MAX_BOXES:
        raise ValueError("amidar state: at most %d boxes on the device engine" % _abi.AMI_MAX_BOXES)
    st.n_boxes = len(boxes)
    for i, bx in enumerate(boxes):
        _strict(bx, BOX_KEYS, "amidar box")
        k = st.boxes[i]
        k.tl_tx, k.tl_ty = int(bx["top_left"]["tx"]), int(bx["top_left"]["ty"])
        k.br_tx, k.br_ty = int(bx["bottom_right"]["tx"]), int(bx["bottom_right"]["ty"])
        k.painted, k.triggers_chase = int(bool(bx["painted"])), int(bool(bx["triggers_chase"]))
    cj = list(b["chase_junctions"])
    if len(cj) > _abi.AMI_MAX_CHASE_J:
        raise ValueError("amidar state: at most %d chase junctions on the device engine" % _abi.AMI_MAX_CHASE_J)
    st.n_chase_junctions = len(cj)
    for i, v in enumerate(cj):
        st.chase_junctions[i] = int(v)
    return st


def schema_for_state():
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "Amidar", "type": "object",
            "required": list(STATE_KEYS), "properties": {}}


def schema_for_config():
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "AmidarConfig", "type": "object",
            "required": list(CONFIG_KEYS), "properties": {}}


def query(tb, name, args):
    """interventions/amidar.py:510,518: 'tile_to_world' {tx,ty} -> [x,y]; 'world_to_tile' {x,y} -> [tx,ty]."""
    if name == "tile_to_world":
        return tb._engine.query(tb._env, _abi.QUERY_TILE_TO_WORLD, [args["tx"], args["ty"]])
    if name == "world_to_tile":
        return tb._engine.query(tb._env, _abi.QUERY_WORLD_TO_TILE, [args["x"], args["y"]])
    js = tb.state_to_json()
    if name == "num_tiles_unpainted":
        return sum(1 for row in js["board"]["tiles"] for t in row if t in ("Unpainted", "ChaseMarker"))
    if name == "jumps_remaining":
        return js["jumps"]
    raise ValueError("unknown amidar query %r" % (name,))
