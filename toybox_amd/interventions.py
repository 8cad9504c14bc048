"""Batched interventions: the reference's Intervention context manager
(/root/reference/toybox/interventions/base.py:371-408) pulls ONE env's JSON, lets the caller mutate it and pushes it
back if dirty.  `BatchIntervention` does that for a whole range of envs with one pack launch + one copy each way:
states come as a numpy structured array (one record per env, fields named like the C structs of
include/toybox_amd.h), so an intervention is a vectorised array expression; per-env JSON views are available for code
written against the interventions schema.
"""
import json
import os

import numpy as np

from .games import codec


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
        self.states = None
        self._before = None
        self._codec = codec(engine.game)
        self.config = None
        self._config_before = None

    def __enter__(self):
        self.states = self.engine.get_states_np(self.first, self.count)
        self._before = self.states.tobytes()
        self.config = self._codec.config_to_json(self.engine.get_config())
        self._config_before = json.dumps(self.config, sort_keys=True)
        return self

    def __exit__(self, exc_type, exc, tb):
        # like Intervention.__exit__ (interventions/base.py:396-406): a changed config is written and a new game started
        # (the config is batch-wide, so for every env of the engine); otherwise changed states are written back
        if exc_type is None:
            if self.dirty_config:
                self.engine.set_config(self._codec.config_from_json(self.config))
                self.engine.new_game()
            elif self.dirty_state:
                self.engine.set_states_np(self.first, self.states)
        self.states = None
        self.config = None
        return False

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
        return self.states is not None and self.states.tobytes() != self._before

    # ---- per-env JSON views (interventions schema) ----
    def json(self, i):
        rec = self.engine.state_type.from_buffer_copy(self.states[i].tobytes())
        return self._codec.state_to_json(rec)

    def write_json(self, i, js):
        rec = self._codec.state_from_json(js)
        self.states[i] = np.frombuffer(bytes(rec), dtype=self.states.dtype)[0]

    # ---- a few vectorised helpers mirroring the reference's intervention classes ----
    def breakout_add_channel(self, col, envs=slice(None)):
        """BreakoutIntervention.add_channel (interventions/breakout.py:392-396) for many envs at once."""
        b = self.states["bricks"][envs]
        sel = (b["col"] == col) & (np.arange(b.shape[-1]) < self.states["n_bricks"][envs][..., None])
        b["alive"][sel] = 0
        self.states["bricks"][envs] = b

    def breakout_bricks_remaining(self):
        b = self.states["bricks"]
        live = (np.arange(b.shape[-1])[None, :] < self.states["n_bricks"][:, None]) & (b["alive"] != 0)
        return live.sum(axis=1)

    def amidar_set_mode(self, mode, time=None, config=None):
        """AmidarIntervention.set_mode (interventions/amidar.py:402-416)."""
        if mode == "jump":
            self.states["jump_timer"] = time or (config or {}).get("jump_time", 75)
        elif mode == "chase":
            self.states["chase_timer"] = time or (config or {}).get("chase_time", 300)
        elif mode == "regular":
            self.states["jump_timer"] = 0
            self.states["chase_timer"] = 0
        else:
            raise ValueError("set_mode not defined for %s" % mode)

    def space_invaders_remove_mothership(self):
        """SpaceInvadersIntervention.remove_mothership (interventions/space_invaders.py:172-173)."""
        self.states["ufo_appearance_counter"] = -1
