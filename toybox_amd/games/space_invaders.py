"""SpaceInvaders: POD records <-> the interventions JSON schema.

State key set == the kwargs of /root/reference/toybox/interventions/space_invaders.py:16-19 with
Player (:38), Laser (:60), Ufo (:101), Enemy (:116), EnemiesMovementState (:146) and
SpriteData (interventions/core.py:224); config keys == the golden dump
toybox/interventions/defaults/space_invaders_config_default.json.  Option<i32> counters are
null <-> -1 in the records.
"""
from .. import _abi
from .._abi import Color, SIConfig, SIState

STATE_KEYS = ["score", "ship_laser", "enemies", "rand", "ufo", "ship", "life_display_timer", "shields",
              "enemies_movement", "lives", "level", "enemy_lasers", "enemy_shot_delay"]
SHIP_KEYS = ["x", "y", "w", "h", "speed", "color", "alive", "death_counter", "death_hit_1"]
LASER_KEYS = ["y", "x", "w", "h", "t", "movement", "speed", "color"]
UFO_KEYS = ["x", "y", "appearance_counter", "death_counter"]
ENEMY_KEYS = ["x", "y", "row", "col", "id", "alive", "points", "death_counter"]
MOVE_KEYS = ["move_counter", "move_dir", "visual_orientation"]
SHIELD_KEYS = ["x", "y", "data"]
CONFIG_KEYS = ["jitter", "shields", "rand", "row_scores", "enemy_protocol", "start_lives"]
PROTOCOLS = {"TargetPlayer": 0}


def _strict(d, keys, what):
    """Required keys must be present; unknown keys are ignored, as serde does on the Rust side (the reference's own
    MovementAI.encode leaks `_in_init` / `schema` into the AI parameters, interventions/amidar.py:159-164)."""
    missing = set(keys) - set(d.keys())
    if missing:
        raise ValueError("%s: missing keys %s" % (what, sorted(missing)))


def _opt(v):
    return None if v < 0 else int(v)


def _unopt(v):
    return -1 if v is None else int(v)


# ------------------------------------------------------------------ config

def config_from_json(d):
    _strict(d, CONFIG_KEYS, "space_invaders config")
    cfg = SIConfig()
    cfg.rand[0], cfg.rand[1] = (int(v) for v in d["rand"]["state"])
    cfg.jitter = float(d["jitter"])
    cfg.start_lives = int(d["start_lives"])
    if d["enemy_protocol"] not in PROTOCOLS:
        raise ValueError("space_invaders config: enemy_protocol %r is not implemented (have %s)" %
                         (d["enemy_protocol"], sorted(PROTOCOLS)))
    cfg.enemy_protocol = PROTOCOLS[d["enemy_protocol"]]
    scores = d["row_scores"]
    if not 1 <= len(scores) <= _abi.SI_MAX_ROWS:
        raise ValueError("space_invaders config: 1..%d enemy rows supported" % _abi.SI_MAX_ROWS)
    cfg.n_rows = len(scores)
    for i, s in enumerate(scores):
        cfg.row_scores[i] = int(s)
    shields = d["shields"]
    if len(shields) > _abi.SI_MAX_SHIELDS:
        raise ValueError("space_invaders config: at most %d shields" % _abi.SI_MAX_SHIELDS)
    cfg.n_shields = len(shields)
    for i, (x, y) in enumerate(shields):
        cfg.shield_x[i], cfg.shield_y[i] = int(x), int(y)
    return cfg


def config_to_json(cfg):
    inv = {v: k for k, v in PROTOCOLS.items()}
    return {
        "jitter": cfg.jitter,
        "shields": [[cfg.shield_x[i], cfg.shield_y[i]] for i in range(cfg.n_shields)],
        "rand": {"state": [int(cfg.rand[0]), int(cfg.rand[1])]},
        "row_scores": [cfg.row_scores[i] for i in range(cfg.n_rows)],
        "enemy_protocol": inv[cfg.enemy_protocol],
        "start_lives": cfg.start_lives,
    }


def default_config():
    return config_from_json({"jitter": 0.5, "shields": [[84, 157], [148, 157], [212, 157]],
                             "rand": {"state": [0x193A6754A8A7D469 ^ 17, 0x97830E05113BA7BB]},
                             "row_scores": [30, 30, 20, 20, 10, 10], "enemy_protocol": "TargetPlayer", "start_lives": 3})


# ------------------------------------------------------------------ state

def _laser_to_json(l):
    return {"y": l.y, "x": l.x, "w": l.w, "h": l.h, "t": l.t, "movement": _abi.DIR_NAMES[l.movement & 3],
            "speed": l.speed, "color": l.color.to_json()}


def _laser_from_json(d, l):
    _strict(d, LASER_KEYS, "laser")
    l.x, l.y, l.w, l.h, l.t, l.speed = (int(d[k]) for k in ("x", "y", "w", "h", "t", "speed"))
    l.movement = _abi.DIR_NAMES.index(d["movement"])
    l.color = Color.from_json(d["color"])


def state_to_json(st):
    blank = {"r": 0, "g": 0, "b": 0, "a": 0}
    shields = []
    for k in range(st.n_shields):
        col = st.shield_color[k].to_json()
        data = [[dict(col) if (st.shield_rows[k][r] >> x) & 1 else dict(blank) for x in range(_abi.SI_SHIELD_W)]
                for r in range(_abi.SI_SHIELD_H)]
        shields.append({"x": st.shield_x[k], "y": st.shield_y[k], "data": data})
    return {
        "score": st.score,
        "ship_laser": _laser_to_json(st.ship_laser) if st.has_ship_laser else None,
        "enemies": [{"x": e.x, "y": e.y, "row": e.row, "col": e.col, "id": e.id, "alive": bool(e.alive),
                     "points": e.points, "death_counter": _opt(e.death_counter)}
                    for e in (st.enemies[i] for i in range(st.n_enemies))],
        "rand": {"state": [int(st.rand[0]), int(st.rand[1])]},
        "ufo": {"x": st.ufo_x, "y": st.ufo_y, "appearance_counter": st.ufo_appearance_counter,
                "death_counter": _opt(st.ufo_death_counter)},
        "ship": {"x": st.ship_x, "y": st.ship_y, "w": st.ship_w, "h": st.ship_h, "speed": st.ship_speed,
                 "color": st.ship_color.to_json(), "alive": bool(st.ship_alive),
                 "death_counter": _opt(st.ship_death_counter), "death_hit_1": bool(st.ship_death_hit_1)},
        "life_display_timer": st.life_display_timer,
        "shields": shields,
        "enemies_movement": {"move_counter": st.move_counter, "move_dir": _abi.DIR_NAMES[st.move_dir & 3],
                             "visual_orientation": bool(st.visual_orientation)},
        "lives": st.lives, "level": st.level,
        "enemy_lasers": [_laser_to_json(st.enemy_lasers[i]) for i in range(st.n_enemy_lasers)],
        "enemy_shot_delay": st.enemy_shot_delay,
    }


def state_from_json(d):
    _strict(d, STATE_KEYS, "space_invaders state")
    st = SIState()
    st.rand[0], st.rand[1] = (int(v) for v in d["rand"]["state"])
    st.score, st.lives, st.level = int(d["score"]), int(d["lives"]), int(d["level"])
    st.life_display_timer, st.enemy_shot_delay = int(d["life_display_timer"]), int(d["enemy_shot_delay"])
    if d["ship_laser"] is not None:
        st.has_ship_laser = 1
        _laser_from_json(d["ship_laser"], st.ship_laser)
    lasers = d["enemy_lasers"]
    if len(lasers) > _abi.SI_MAX_LASERS:
        raise ValueError("space_invaders state: at most %d enemy lasers on the device engine" % _abi.SI_MAX_LASERS)
    st.n_enemy_lasers = len(lasers)
    for i, l in enumerate(lasers):
        _laser_from_json(l, st.enemy_lasers[i])
    enemies = d["enemies"]
    if len(enemies) > _abi.SI_MAX_ENEMIES:
        raise ValueError("space_invaders state: at most %d enemies on the device engine" % _abi.SI_MAX_ENEMIES)
    st.n_enemies = len(enemies)
    for i, e in enumerate(enemies):
        _strict(e, ENEMY_KEYS, "enemy")
        k = st.enemies[i]
        k.x, k.y, k.row, k.col, k.id, k.points = (int(e[f]) for f in ("x", "y", "row", "col", "id", "points"))
        k.alive = int(bool(e["alive"]))
        k.death_counter = _unopt(e["death_counter"])
    u = d["ufo"]
    _strict(u, UFO_KEYS, "ufo")
    st.ufo_x, st.ufo_y, st.ufo_appearance_counter = int(u["x"]), int(u["y"]), int(u["appearance_counter"])
    st.ufo_death_counter = _unopt(u["death_counter"])
    s = d["ship"]
    _strict(s, SHIP_KEYS, "ship")
    st.ship_x, st.ship_y, st.ship_w, st.ship_h, st.ship_speed = (int(s[f]) for f in ("x", "y", "w", "h", "speed"))
    st.ship_color = Color.from_json(s["color"])
    st.ship_alive, st.ship_death_hit_1 = int(bool(s["alive"])), int(bool(s["death_hit_1"]))
    st.ship_death_counter = _unopt(s["death_counter"])
    m = d["enemies_movement"]
    _strict(m, MOVE_KEYS, "enemies_movement")
    st.move_counter, st.move_dir = int(m["move_counter"]), _abi.DIR_NAMES.index(m["move_dir"])
    st.visual_orientation = int(bool(m["visual_orientation"]))
    shields = d["shields"]
    if len(shields) > _abi.SI_MAX_SHIELDS:
        raise ValueError("space_invaders state: at most %d shields on the device engine" % _abi.SI_MAX_SHIELDS)
    st.n_shields = len(shields)
    for k, sh in enumerate(shields):
        _strict(sh, SHIELD_KEYS, "shield")
        st.shield_x[k], st.shield_y[k] = int(sh["x"]), int(sh["y"])
        data = sh["data"]
        if len(data) != _abi.SI_SHIELD_H or any(len(row) != _abi.SI_SHIELD_W for row in data):
            raise ValueError("space_invaders state: shields are %dx%d sprites" % (_abi.SI_SHIELD_W, _abi.SI_SHIELD_H))
        colour = None
        for r, row in enumerate(data):
            bits = 0
            for x, px in enumerate(row):
                if int(px["a"]) != 0:
                    c = (int(px["r"]), int(px["g"]), int(px["b"]), int(px["a"]))
                    if colour is None:
                        colour = c
                    elif c != colour:
                        raise ValueError("space_invaders state: the device engine keeps one colour per shield")
                    bits |= 1 << x
            st.shield_rows[k][r] = bits
        if colour is None:
            colour = (172, 80, 48, 255)
        st.shield_color[k] = Color(*colour)
    return st


def schema_for_state():
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "SpaceInvaders", "type": "object",
            "required": list(STATE_KEYS), "properties": {}}


def schema_for_config():
    return {"$schema": "http://json-schema.org/draft-07/schema#", "title": "SpaceInvadersConfig", "type": "object",
            "required": list(CONFIG_KEYS), "properties": {}}


def query(tb, name, args):
    js = tb.state_to_json()
    if name == "enemies_remaining":
        return sum(1 for e in js["enemies"] if e["alive"])
    if name == "ship_x":
        return js["ship"]["x"]
    if name == "shield_count":
        return len(js["shields"])
    raise ValueError("unknown space_invaders query %r" % (name,))
