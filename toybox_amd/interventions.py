"""Batched interventions: the reference's Intervention context manager
(/root/reference/toybox/interventions/base.py:371-408) pulls ONE env's JSON, lets the caller mutate it and pushes it
back if dirty.  `BatchIntervention` does that for a whole range of envs with one pack launch + one copy each way:
states come as a numpy structured array (one record per env, fields named like the C structs of
include/toybox_amd.h), so an intervention is a vectorised array expression; per-env JSON views are available for code
written against the interventions schema.
"""
import numpy as np

from .games import codec


class BatchIntervention:
    def __init__(self, engine, first=0, count=None):
        self.engine = engine
        self.first = int(first)
        self.count = engine.n_envs - self.first if count is None else int(count)
        self.states = None
        self._before = None
        self._codec = codec(engine.game)

    def __enter__(self):
        self.states = self.engine.get_states_np(self.first, self.count)
        self._before = self.states.tobytes()
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None and self.dirty_state:
            self.engine.set_states_np(self.first, self.states)
        self.states = None
        return False

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
