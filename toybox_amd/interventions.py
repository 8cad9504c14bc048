"""Batched interventions.

The reference's `Intervention` context manager (/root/reference/toybox/interventions/base.py:371-408) pulls ONE env's JSON,
lets the caller mutate it and pushes it back if dirty; its per-game subclasses add helper methods over that one decoded
state (interventions/breakout.py:303-429, amidar.py:360-615, space_invaders.py:165-176).  `BatchIntervention` is the same
thing over a whole engine:

* the helper methods, under the reference's names, as kernels over the state in HBM (`Engine.edit` / `Engine.reduce` ->
  tbx_edit / tbx_reduce): an edit takes `envs=` (a boolean mask) and scalar arguments or one value per env, a query returns
  one value (or row) per env -- a 65 536-env sweep never copies a state record to the host;
* `.states`, the numpy structured array of every env's POD record (fields named like the C structs of
  include/toybox_amd.h), fetched on first access and written back on exit if it was changed -- for edits no helper covers;
* `.config`, the batch-wide config JSON (`dirty_config` -> tbx_set_config + new game, like interventions/base.py:396-403),
  `set_partial_config`, `differs` / `diff_states` (the batch form of `SetEq`), per-env JSON views.

Helpers of the reference that only draw random tiles / directions (get_random_tile, get_random_track_position,
get_random_dir_for_tile, the draw inside set_player_random_start) are host-side tooling around a predicate and have no
batched form; their deterministic halves do (`set_player_tile`, `count_tiles`, `get_adjacent_tiles`).
"""
import json
import os

import numpy as np

from . import _abi
from .games import codec

TILE_TAGS = ["Empty", "Unpainted", "Painted", "ChaseMarker"]       # interventions/amidar.py:55-59; index == TBX_TILE_*
TILE_WX, TILE_WY = 64, 80                                         # world units per tile (TBX_AMI_TILE_WX / WY)
BOARD_W, BOARD_H = 32, 31                                         # TBX_AMI_BOARD_W / H
DIRECTIONS = ["Up", "Down", "Left", "Right"]                      # interventions/core.py:125-135; index == TBX_DIR_*


def diff_states(a, b, rel_tol=1e-9, prefix=""):
    """SetEq for batches (interventions/base.py:47-106): where do two arrays of state records differ?  Returns
    [(field path, env indices)], floats compared like math.isclose (rel_tol 1e-9), everything else exactly; nested
    records and per-entity arrays are walked by name, e.g. 'bricks.alive' or 'ball_x'."""
    assert a.dtype == b.dtype and a.shape == b.shape
    out = []
    for name in a.dtype.names:
        if name.startswith("_"):
            continue
        x, y = a[name], b[name]
        path = prefix + name
        if x.dtype.names:
            out.extend(diff_states(x, y, rel_tol, path + "."))
            continue
        if x.dtype.kind == "f":
            neq = ~np.isclose(x, y, rtol=rel_tol, atol=0.0, equal_nan=True)
        else:
            neq = x != y
        envs = np.flatnonzero(neq.reshape(neq.shape[0], -1).any(axis=1))
        if len(envs):
            out.append((path, envs))
    return out


class BatchIntervention:
    def __init__(self, engine, first=0, count=None):
        self.engine = engine
        self.first = int(first)
        self.count = engine.n_envs - self.first if count is None else int(count)
        self._states = None
        self._before = None
        self._codec = codec(engine.game)
        self.config = None
        self._config_before = None
        self._active = False

    def __enter__(self):
        self._states = None                         # fetched on first use: the helper methods never need the records
        self._before = None
        self.config = self._codec.config_to_json(self.engine.get_config())
        self._config_before = json.dumps(self.config, sort_keys=True)
        self._active = True
        return self

    def __exit__(self, exc_type, exc, tb):
        # like Intervention.__exit__ (interventions/base.py:396-406): a changed config is written and a new game started
        # (the config is batch-wide, so for every env of the engine); otherwise changed states are written back
        if exc_type is None:
            if self.dirty_config:
                self.engine.set_config(self._codec.config_from_json(self.config))
                self.engine.new_game()
            elif self.dirty_state:
                self.engine.set_states_np(self.first, self._states)
        self._states = None
        self.config = None
        self._active = False
        return False

    # ---- the records on the host (only when asked for)
    @property
    def states(self):
        if self._states is None and self._active:
            self._states = self.engine.get_states_np(self.first, self.count)
            self._before = self._states.tobytes()
        return self._states

    def _flush(self):
        """a device-side helper is about to run: records edited on the host go down first, and the host copy is dropped.
        An array a caller took from `.states` before this point is STALE afterwards (the device state has moved on): it is made
        read-only, so that a later write through it raises instead of being lost -- read `.states` again."""
        if self.dirty_state:
            self.engine.set_states_np(self.first, self._states)
        if self._states is not None:
            self._states.flags.writeable = False
        self._states = None
        self._before = None

    @property
    def dirty_config(self):
        return self.config is not None and json.dumps(self.config, sort_keys=True) != self._config_before

    def set_partial_config(self, source):
        """Intervention.set_partial_config (interventions/base.py:409-419): keys of a JSON file (or a dict) that the config
        has replace its values."""
        if isinstance(source, dict):
            data = source
        elif os.path.isfile(source):
            with open(source) as f:
                data = json.load(f)
        else:
            return
        for k, v in data.items():
            if k in self.config:
                self.config[k] = v

    def differs(self, other_states, rel_tol=1e-9):
        """SetEq-style report against another batch of records (see diff_states)."""
        return diff_states(self.states, other_states, rel_tol)

    @property
    def dirty_state(self):
        return self._states is not None and self._states.tobytes() != self._before

    # ---- per-env JSON views (interventions schema) ----
    def json(self, i):
        rec = self.engine.state_type.from_buffer_copy(self.states[i].tobytes())
        return self._codec.state_to_json(rec)

    def write_json(self, i, js):
        rec = self._codec.state_from_json(js)
        self.states[i] = np.frombuffer(bytes(rec), dtype=self.states.dtype)[0]

    # ---- plumbing of the device-side helpers
    def _mask(self, envs):
        """envs: None (every env of the range), a boolean mask over the range, or indices into it -> mask over the engine"""
        n = self.engine.n_envs
        m = np.zeros(n, bool)
        if envs is None or (isinstance(envs, slice) and envs == slice(None)):
            m[self.first:self.first + self.count] = True
            return None if self.count == n else m
        sel = np.zeros(self.count, bool)
        sel[envs] = True
        m[self.first:self.first + self.count] = sel
        return m

    def _args(self, *cols):
        """scalars -> one row for every env; any array-valued column -> one row per env of the engine"""
        if not any(np.ndim(c) for c in cols):
            return [float(c) for c in cols]
        out = np.zeros((self.engine.n_envs, len(cols)), np.float64)
        for k, c in enumerate(cols):
            if np.ndim(c):
                out[self.first:self.first + self.count, k] = np.asarray(c, np.float64)
            else:
                out[:, k] = float(c)
        return out

    def _edit(self, op, *cols, envs=None):
        self._flush()
        self.engine.edit(op, self._args(*cols), self._mask(envs))

    def _reduce(self, query, *cols):
        self._flush()
        return self.engine.reduce(query, self._args(*cols))[self.first:self.first + self.count]

    def _need(self, game):
        if self.engine.game != game:
            raise TypeError("this helper belongs to %s interventions, the engine plays %s" % (game, self.engine.game))

    # ---- every game: intervention.game.lives = v (interventions/space_invaders.py:199)
    def set_lives(self, lives, envs=None):
        self._edit(_abi.EDIT_SET_LIVES, lives, envs=envs)

    def set_score(self, score, envs=None):
        self._edit(_abi.EDIT_SET_SCORE, score, envs=envs)

    def set_level(self, level, envs=None):
        self._edit(_abi.EDIT_SET_LEVEL, level, envs=envs)

    # ================================================================== BreakoutIntervention (interventions/breakout.py)
    def num_bricks_remaining(self):
        """:309-310 -> int[N]"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_BRICKS_REMAINING)[:, 0].astype(np.int64)

    def num_bricks(self):
        """:312-313"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_NUM_BRICKS)[:, 0].astype(np.int64)

    def num_rows(self):
        """:315-316 (batch-wide: the config's row_scores)"""
        self._need("breakout")
        return len(self.config["row_scores"])

    def num_columns(self):
        """:318-322 -> int[N]: bricks // rows"""
        return self.num_bricks() // self.num_rows()

    def get_column(self, i):
        """:349-355 -> int[N, k]: the alive flags of the bricks of column i in brick order (k = the most any env has; -1 pads)"""
        self._need("breakout")
        return self._trim(self._reduce(_abi.QUERY_BRK_COLUMN, i))

    def get_row(self, i):
        """:357-359 as documented (the ith row; the reference's body compares with 1)"""
        self._need("breakout")
        return self._trim(self._reduce(_abi.QUERY_BRK_ROW, i))

    @staticmethod
    def _trim(a):
        a = a.astype(np.int64)
        used = (a >= 0).any(axis=0)
        return a[:, :int(used.sum())] if used.any() else a[:, :0]

    def is_channel(self, col):
        """:340-347 for the bricks of one column -> bool[N]"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_IS_CHANNEL, col)[:, 0] != 0

    def is_stack(self, col):
        """:336-338: the bricks get_column returns share their column by construction"""
        self._need("breakout")
        return np.ones(self.count, bool)

    def channel_count(self):
        """:361-366 -> int[N]"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_CHANNEL_COUNT)[:, 0].astype(np.int64)

    def find_channel(self):
        """:404-410 -> int[N]: the first channel's column, -1 without one"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_FIND_CHANNEL)[:, 0].astype(np.int64)

    def brick_table(self, env=0):
        """the attributes of the bricks that a predicate may look at, as the reference's Brick objects carry them (:196-222), from
        env `env` of the range: everything but `alive` is the same in every env of a batch that shares its config"""
        from types import SimpleNamespace
        js = self.json(env)
        out = []
        for b in js["bricks"]:
            ns = SimpleNamespace(**{k: v for k, v in b.items() if k not in ("position", "size", "color")})
            ns.position = SimpleNamespace(**b["position"]); ns.size = SimpleNamespace(**b["size"]); ns.color = SimpleNamespace(**b["color"])
            out.append(ns)
        return out

    def find_brick(self, pred, alive=None):
        """:400-404 over the batch -> int[N]: the index of the first brick that satisfies the predicate, -1 where none does (the
        reference raises ValueError for its one env).  `pred` is either a boolean mask over brick indices or a Python predicate
        over a brick's STATIC attributes (row, col, points, depth, color, position, size, destructible): it is evaluated ONCE over
        the brick table -- those attributes are the same in every env -- and turned into the mask of TBX_QUERY_BRK_FIND_BRICK.
        The one per-env attribute, `alive`, is given beside it: alive=True / False restricts the search, None takes either.
        (A predicate that reads b.alive sees env 0's flags: pass alive= instead.)"""
        self._need("breakout")
        if callable(pred):
            mask = np.array([bool(pred(b)) for b in self.brick_table()], bool)
        else:
            mask = np.asarray(pred, bool)
        if mask.ndim != 1 or len(mask) > _abi.BRK_MAX_BRICKS:
            raise ValueError("the brick mask must be one-dimensional with at most %d entries" % _abi.BRK_MAX_BRICKS)
        bits = np.zeros(8 * 32, np.uint64)
        bits[:len(mask)] = mask
        words = [int((bits[32 * k:32 * k + 32] << np.arange(32, dtype=np.uint64)).sum()) for k in range(8)]
        want = -1 if alive is None else int(bool(alive))
        return self._reduce(_abi.QUERY_BRK_FIND_BRICK, want, *words)[:, 0].astype(np.int64)

    def add_channel(self, i, envs=None):
        """:392-396: turns column i into a channel (i may be an array: env k gets column i[k])"""
        self._need("breakout")
        self._edit(_abi.EDIT_BRK_COLUMN_ALIVE, i, 0, envs=envs)

    def fill_column(self, i, envs=None):
        """:398-402"""
        self._need("breakout")
        self._edit(_abi.EDIT_BRK_COLUMN_ALIVE, i, 1, envs=envs)

    def clear_board(self, envs=None):
        """:412-415"""
        self._need("breakout")
        self._edit(_abi.EDIT_BRK_ALL_ALIVE, 0, envs=envs)

    def set_row_alive(self, row, alive, envs=None):
        self._need("breakout")
        self._edit(_abi.EDIT_BRK_ROW_ALIVE, row, int(bool(alive)), envs=envs)

    def set_brick_alive(self, index, alive, envs=None):
        """bricks[index].alive = alive (test_breakout_interventions.py:48)"""
        self._need("breakout")
        self._edit(_abi.EDIT_BRK_BRICK_ALIVE, index, alive, envs=envs)

    def get_paddle_position(self):
        """:379-380 -> float64[N, 2]"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_PADDLE)[:, 0:2]

    def get_paddle_velocity(self):
        """:382-383"""
        self._need("breakout")
        return self._reduce(_abi.QUERY_BRK_PADDLE)[:, 2:4]

    def set_paddle_position(self, x, y=None, envs=None):
        self._need("breakout")
        if y is None:
            self._edit(_abi.EDIT_BRK_PADDLE, x, envs=envs)
        else:
            self._edit(_abi.EDIT_BRK_PADDLE, x, y, envs=envs)

    def _balls(self):
        b = self._reduce(_abi.QUERY_BRK_BALLS)
        return b[:, 0].astype(np.int64), b[:, 1:].reshape(self.count, 4, _abi.BRK_MAX_BALLS)

    def get_ball_position(self):
        """:365-371 -> (n_balls int[N], position float64[N, 4, 2]; absent balls read -1)"""
        self._need("breakout")
        n, b = self._balls()
        return n, np.stack([b[:, 0], b[:, 1]], axis=-1)

    def get_ball_velocity(self):
        """:373-377"""
        self._need("breakout")
        n, b = self._balls()
        return n, np.stack([b[:, 2], b[:, 3]], axis=-1)

    def set_ball(self, ball, x, y, vx, vy, envs=None):
        self._need("breakout")
        self._edit(_abi.EDIT_BRK_BALL, ball, x, y, vx, vy, envs=envs)

    def add_row(self, points, color=None):
        """:324-334 as a config intervention: one more brick row with `points` (every env starts a new game on exit, like the
        reference's dirty_config)"""
        self._need("breakout")
        self.config["row_scores"].append(int(points))
        self.config["row_colors"].append(dict(color or self.config["row_colors"][-1]))

    # the round-3 names, kept
    def breakout_add_channel(self, col, envs=None):
        self.add_channel(col, envs=None if isinstance(envs, slice) and envs == slice(None) else envs)

    def breakout_bricks_remaining(self):
        return self.num_bricks_remaining()

    # ================================================================== AmidarIntervention (interventions/amidar.py)
    def _mode(self):
        self._need("amidar")
        m = self._reduce(_abi.QUERY_AMI_MODE).astype(np.int64)
        return m[:, 0], m[:, 1]

    def get_regular_mode(self):
        """:385-387"""
        j, c = self._mode()
        return (j == 0) & (c == 0)

    def get_jump_mode(self):
        """:389-391"""
        return self._mode()[0] > 0

    def get_chase_mode(self):
        """:393-395"""
        return self._mode()[1] > 0

    def any_enemy_caught(self):
        """:397-399"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_ANY_CAUGHT)[:, 0] != 0

    def set_mode(self, mode, set_time=None, envs=None):
        """:402-416"""
        self._need("amidar")
        def timer(default):
            # the reference: `set_time or config[...]` -- a zero (or missing) time means the config's; per env for an array
            if set_time is None:
                return default
            if np.ndim(set_time) > 0:
                t = np.asarray(set_time)
                return np.where(t == 0, default, t)
            return set_time if set_time else default
        if mode == "jump":
            self._edit(_abi.EDIT_AMI_TIMERS, timer(self.config["jump_time"]), -1, envs=envs)
        elif mode == "chase":
            self._edit(_abi.EDIT_AMI_TIMERS, -1, timer(self.config["chase_time"]), envs=envs)
        elif mode == "regular":
            self._edit(_abi.EDIT_AMI_TIMERS, 0, 0, envs=envs)
        else:
            raise ValueError("set_mode not defined for %s" % mode)

    def amidar_set_mode(self, mode, time=None, config=None):
        self.set_mode(mode, time or (config or {}).get({"jump": "jump_time", "chase": "chase_time"}.get(mode, ""), None))

    def set_jumps(self, jumps, envs=None):
        self._need("amidar")
        self._edit(_abi.EDIT_AMI_JUMPS, jumps, envs=envs)

    def set_enemy_protocol(self, enemy, protocol, envs=None, **kwargs):
        """:418-471: enemy = index into game.enemies; kwargs are the protocol's parameters in JSON form (tile points as
        {'tx','ty'}, directions by name), exactly what the reference leaves in the state after the call"""
        self._need("amidar")
        from .games import amidar as am
        ai = am.ai_from_json({protocol: kwargs})
        self._edit(_abi.EDIT_AMI_ENEMY_AI, enemy, *[getattr(ai, f) for f, _ in _abi.AmidarAI._fields_], envs=envs)

    def get_tile_by_pos(self, tx, ty):
        """:480-481 -> tag names, object array [N] (None outside the board)"""
        self._need("amidar")
        t = self._reduce(_abi.QUERY_AMI_TILE, tx, ty)[:, 0].astype(np.int64)
        return np.array([TILE_TAGS[v] if v >= 0 else None for v in t], dtype=object)

    def is_tile_walkable(self, tx, ty):
        """:472-474"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_TILE, tx, ty)[:, 0] > 0

    def set_tile_tag(self, tx, ty, tag, envs=None):
        """:476-478"""
        self._need("amidar")
        assert tag in TILE_TAGS, "Unrecognized tile tag: %s" % tag
        self._edit(_abi.EDIT_AMI_TILE, tx, ty, TILE_TAGS.index(tag), envs=envs)

    def count_tiles(self, tag):
        """len(filter_tiles(lambda t: t.tag == tag)) :483-488 -> int[N]"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_COUNT_TILES, TILE_TAGS.index(tag))[:, 0].astype(np.int64)

    @staticmethod
    def _tag_mask(tags):
        """tag names (or a predicate over a tag name) -> bit mask over TILE_TAGS"""
        if callable(tags):
            return sum(1 << i for i, t in enumerate(TILE_TAGS) if tags(t))
        if isinstance(tags, str):
            tags = [tags]
        return sum(1 << TILE_TAGS.index(t) for t in tags)

    def filter_tiles(self, tags=lambda tag: True):
        """:494-499 for predicates on the tile's tag -> bool[N, 31, 32]: [env, ty, tx] is set where the reference's list would
        hold tile (tx, ty).  `tags`: a tag name, a collection of them, or a predicate over the tag name (evaluated once for
        each of the four tags).  np.argwhere(result[env]) lists the tiles in the reference's row-by-row order."""
        self._need("amidar")
        rows = self._reduce(_abi.QUERY_AMI_TILES_MASK, self._tag_mask(tags))[:, :BOARD_H].astype(np.uint64)
        return ((rows[:, :, None] >> np.arange(BOARD_W, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool)

    def count_filtered_tiles(self, tags):
        """len(filter_tiles(pred on the tag)) -> int[N]"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_TILES_MASK, self._tag_mask(tags))[:, 31].astype(np.int64)

    @staticmethod
    def tile_to_tilepoint(tx, ty):
        """:501-506: in the reference a Tile object is looked up by identity to find its (tx, ty); a batched tile IS its (tx, ty)"""
        return np.asarray(tx), np.asarray(ty)

    @classmethod
    def tile_to_worldpoint(cls, tx, ty):
        """:512-514"""
        return cls.tilepoint_to_worldpoint(tx, ty)

    @staticmethod
    def _counter_args(seed, draw, env_offset):
        """The counter rule's three arguments travel as binary64 and are read as unsigned 32-bit integers on the device
        (TbxEditArgs::getu saturates above that: every larger seed would name ONE stream).  A seed of any size -- the
        64-bit hash_seed() of the gym layer, say -- is folded to 32 bits the way gym folds its own (sha-free: xor of the halves);
        draw and env_offset are counters and must fit as they are."""
        def fold(v):
            v = int(v)
            if v < 0:
                raise ValueError("seed must be >= 0")
            while v >> 32:
                v = (v & 0xFFFFFFFF) ^ (v >> 32)
            return v
        seed = fold(seed) if np.ndim(seed) == 0 else np.asarray([fold(v) for v in np.asarray(seed).ravel()], np.float64)
        out = [seed]
        for name, v in (("draw", draw), ("env_offset", env_offset)):       # (one value for the batch, or one per env)
            a = np.asarray(v)
            if a.size and (a.min() < 0 or a.max() >= 1 << 32):
                raise ValueError("%s must be in [0, 2**32)" % name)
            out.append(int(v) if a.ndim == 0 else a)
        return tuple(out)

    def get_random_tile(self, tags=lambda tag: True, seed=0, draw=0, env_offset=0, min_enemy_distance=0):
        """:360-378 with the counter rule instead of `random` (include/toybox_amd.h): draw number `draw` of env e picks element
        splitmix64(seed ^ (env_offset + e) << 32 ^ draw) mod len of the list filter_tiles(pred) would return for that env --
        uniform over exactly the tiles the reference's rejection loop can return, reproducible, not `random`-compatible.
        pred: the tag is in `tags` and (min_enemy_distance > 0) set_player_random_start's within_min_manhattan.
        -> (tx int[N], ty int[N], tag object[N], candidates int[N]); -1 / None where an env has no candidate (the reference raises)."""
        self._need("amidar")
        seed, draw, env_offset = self._counter_args(seed, draw, env_offset)
        r = self._reduce(_abi.QUERY_AMI_RANDOM_TILE, seed, draw, env_offset, self._tag_mask(tags), min_enemy_distance).astype(np.int64)
        return r[:, 0], r[:, 1], np.array([TILE_TAGS[v] if v >= 0 else None for v in r[:, 2]], dtype=object), r[:, 3]

    def get_random_track_position(self, seed=0, draw=0, env_offset=0):
        """:380-386 -> world (x int[N], y int[N]) of a random tile whose tag is not Empty"""
        tx, ty, _, _ = self.get_random_tile(lambda tag: tag != "Empty", seed, draw, env_offset)
        x, y = self.tilepoint_to_worldpoint(tx, ty)
        return np.where(tx >= 0, x, -1), np.where(ty >= 0, y, -1)

    def set_player_random_start(self, min_enemy_distance=5, seed=0, draw=0, env_offset=0, envs=None):
        """:541-548 on the device: the player of every selected env goes to get_random_tile(within_min_manhattan) -- ANY tile, as
        the reference draws it (its predicate does not ask for a walkable one), for which not every enemy is nearer than
        min_enemy_distance"""
        self._need("amidar")
        seed, draw, env_offset = self._counter_args(seed, draw, env_offset)
        self._edit(_abi.EDIT_AMI_PLAYER_RANDOM_START, seed, draw, env_offset, min_enemy_distance, envs=envs)

    def get_random_dir_for_tile(self, tx, ty, seed=0, draw=0, env_offset=0):
        """:550-583 -> direction names, object array [N]: drawn (counter rule) among those of Up, Down, Left, Right whose neighbour
        tile is walkable; None where there is none (the reference raises)"""
        self._need("amidar")
        seed, draw, env_offset = self._counter_args(seed, draw, env_offset)
        r = self._reduce(_abi.QUERY_AMI_RANDOM_DIR, seed, draw, env_offset, tx, ty).astype(np.int64)
        return np.array([DIRECTIONS[v] if v >= 0 else None for v in r[:, 0]], dtype=object)

    @staticmethod
    def tilepoint_to_worldpoint(tx, ty):
        """:497-499 (the engine's tile_to_world query: pure arithmetic, vectorised here)"""
        return np.asarray(tx) * TILE_WX, np.asarray(ty) * TILE_WY

    @staticmethod
    def worldpoint_to_tilepoint(x, y):
        """:505-507"""
        return np.floor_divide(np.asarray(x), TILE_WX), np.floor_divide(np.asarray(y), TILE_WY)

    def get_adjacent_tiles(self, tx, ty):
        """:509-524 -> int[N, 4]: tile tags (index into TILE_TAGS) of the up, left, right and down neighbours, the order in
        which the reference's row-major filter_tiles scan meets them; -1 outside the board"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_ADJACENT, tx, ty).astype(np.int64)

    def enemy_distances_from_tile(self, tx, ty):
        """:526-530 with TilePoint.manhattan -> int[N, 8] (-1 for absent enemies)"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_ENEMY_DISTANCES, tx, ty).astype(np.int64)

    def set_player_tile(self, tx, ty, envs=None):
        """the write at the end of set_player_random_start (:531-538): player.position = tile_to_worldpoint(tile); the
        caller draws the tile (per env if it likes) -- e.g. with enemy_distances_from_tile as the reference's predicate"""
        self._need("amidar")
        self._edit(_abi.EDIT_AMI_PLAYER_TILE, tx, ty, envs=envs)

    def player_tile(self):
        """:573-576 -> (tx int[N], ty int[N], tag object[N])"""
        self._need("amidar")
        p = self._reduce(_abi.QUERY_AMI_PLAYER_TILE).astype(np.int64)
        return p[:, 0], p[:, 1], np.array([TILE_TAGS[v] if v >= 0 else None for v in p[:, 2]], dtype=object)

    def player_enemy_distances(self):
        """:579-583"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_PLAYER_ENEMY_DISTANCES).astype(np.int64)

    def player_on_painted(self):
        """:586-589"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_PLAYER_ON_PAINTED)[:, 0] != 0

    def player_near_unpainted(self, radius=5):
        """:592-603"""
        self._need("amidar")
        return self._reduce(_abi.QUERY_AMI_PLAYER_NEAR_UNPAINTED, radius)[:, 0] != 0

    # ================================================================== SpaceInvadersIntervention (interventions/space_invaders.py)
    def get_jitter(self):
        """:165-166"""
        self._need("space_invaders")
        return self.config["jitter"]

    def set_jitter(self, p):
        """:168-170 (a config intervention: written, and a new game started, on exit)"""
        self._need("space_invaders")
        self.config["jitter"] = p

    def remove_mothership(self, banish_time=None, envs=None):
        """:172-173"""
        self._need("space_invaders")
        self._edit(_abi.EDIT_SI_UFO_APPEARANCE, -1, envs=envs)

    def space_invaders_remove_mothership(self):
        self.remove_mothership()

    def get_player(self):
        """:175-176 -> dict of int arrays [N]: the ship's fields"""
        self._need("space_invaders")
        s = self._reduce(_abi.QUERY_SI_SHIP).astype(np.int64)
        keys = ["x", "y", "w", "h", "speed", "alive", "death_counter", "death_hit_1"]
        out = {k: s[:, i] for i, k in enumerate(keys)}
        out["alive"] = out["alive"] != 0
        out["death_hit_1"] = out["death_hit_1"] != 0
        return out
