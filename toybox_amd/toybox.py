"""ctoybox-shaped single-env API over the batched engine.

`Toybox`, `Input`, `Simulator`, `State` keep the method names, argument meaning and error behaviour
that the reference's Python code uses from the `ctoybox` module (call sites:
/root/reference/toybox/envs/atari/base.py:41-167, toybox/interventions/base.py:371-408,
scripts/utils/test_games.py:5-41, test/benchmark.py:44-58, scripts/utils/start_images_toybox:24-37).
A `Toybox` is a view of ONE env of an `Engine` (by default an engine of its own with n_envs=1), so
all arithmetic still happens in the HIP kernels behind the C-ABI.
"""
import json
import struct
import zlib

import numpy as np

from . import _abi
from .engine import Engine
from .games import codec

GAME_ALIASES = {"breakout": "breakout", "amidar": "amidar", "space_invaders": "space_invaders",
                "spaceinvaders": "space_invaders", "gridworld": "gridworld"}

_engine_factory = None


def set_engine_factory(fn):
    """fn(game_name, n_envs) -> Engine.  None restores the default (the in-tree HIP library)."""
    global _engine_factory
    _engine_factory = fn


def _make_engine(game, n_envs=1):
    if _engine_factory is not None:
        return _engine_factory(game, n_envs)
    return Engine(game, n_envs)


class Input(object):
    """Button state of one frame (ctoybox.Input)."""
    _LEFT = "left"
    _RIGHT = "right"
    _UP = "up"
    _DOWN = "down"
    _BUTTON1 = "button1"
    _BUTTON2 = "button2"
    _NOOP = "noop"

    def __init__(self):
        self.reset()

    def reset(self):
        self.left = False
        self.right = False
        self.up = False
        self.down = False
        self.button1 = False
        self.button2 = False

    def __str__(self):
        return self.__repr__()

    def __repr__(self):
        return "<ctoybox.Input left={0.left} right={0.right} up={0.up} down={0.down} " \
               "button1={0.button1} button2={0.button2}>".format(self)

    def set_input(self, input_dir, button=_NOOP):
        input_dir, button = input_dir.lower(), button.lower()
        if input_dir == Input._NOOP:
            pass
        elif input_dir in (Input._LEFT, Input._RIGHT, Input._UP, Input._DOWN):
            setattr(self, input_dir, True)
        else:
            raise ValueError("input_dir must be one of noop/left/right/up/down, not %r" % input_dir)
        if button == Input._NOOP:
            pass
        elif button in (Input._BUTTON1, Input._BUTTON2):
            setattr(self, button, True)
        else:
            raise ValueError("button must be one of noop/button1/button2, not %r" % button)

    def to_mask(self):
        m = 0
        if self.left: m |= _abi.BTN_LEFT
        if self.right: m |= _abi.BTN_RIGHT
        if self.up: m |= _abi.BTN_UP
        if self.down: m |= _abi.BTN_DOWN
        if self.button1: m |= _abi.BTN_BUTTON1
        if self.button2: m |= _abi.BTN_BUTTON2
        return m


def write_png(path, frame):
    """Minimal PNG writer (8-bit gray / RGB / RGBA) -- Toybox.save_frame_image needs no imaging library."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    if frame.ndim == 2:
        frame = frame[:, :, None]
    h, w, c = frame.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + frame[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


class Toybox(object):
    """One game env with the ctoybox.Toybox surface."""

    def __init__(self, game_name, grayscale=True, frameskip=0, seed=None, withstate=None, engine=None, env_index=0):
        if game_name not in GAME_ALIASES:
            raise ValueError("unknown game %r" % (game_name,))
        self.game_name = game_name
        self._game = GAME_ALIASES[game_name]
        self._codec = codec(self._game)
        self.frames_per_action = frameskip + 1
        self.grayscale = grayscale
        self._own_engine = engine is None
        self._engine = engine if engine is not None else _make_engine(self._game, 1)   # plays the first new game
        self._env = int(env_index)
        self.rsimulator = Simulator(self)
        self.rstate = State(self)
        self._scal = None          # (score, lives) as of the last apply_ale_action, until anything else touches the state
        if self._own_engine:
            if seed is not None:
                self.set_seed(seed)
            # ctoybox's constructor starts a second game after optional seeding (this is what makes the
            # golden dumps' RNG words come out: tests/golden/rng_kat.json, "children_before_config": 2)
            self.new_game()
            if withstate:
                self.write_state_json(withstate)

    # ------------------------------------------------------------------ lifecycle
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()

    def close(self):
        if self._own_engine and self._engine is not None:
            self._engine.close()
        self._engine = None
        self.rstate = None
        self.rsimulator = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ T: transition
    def new_game(self):
        self._scal = None
        mask = None
        if self._engine.n_envs > 1:
            mask = np.zeros(self._engine.n_envs, np.uint8)
            mask[self._env] = 1
        self._engine.new_game(mask)

    def set_seed(self, seed):
        """Re-seeds the simulator RNG; takes effect at the next new_game() (envs/atari/base.py:95-97)."""
        self._engine.seed(int(seed), env=self._env)

    def get_legal_action_set(self):
        return sorted(self._engine.legal_actions)

    def apply_ale_action(self, action_int):
        if int(action_int) not in self._engine.legal_actions:
            raise ValueError("Expected to apply action, but failed: {0}".format(action_int))
        # one round trip per frame: the step call hands back score and lives, so the get_score / get_lives / game_over
        # calls that follow every action in the reference's loops (test/benchmark.py:50-56) cost nothing.  A one-env engine
        # answers from its resident step kernel (tbx_step1): no launch, no copy, no synchronisation.
        if self._engine.n_envs == 1:
            step1 = self._engine.step1
            for _ in range(self.frames_per_action):
                _, _, lives, score = step1(0, action_int)
            self._scal = (score, lives)
            return
        self._scal = None
        buttons = self._engine._lib.tbx_ale_action_to_buttons(int(action_int))
        for _ in range(self.frames_per_action):
            self._engine.apply_input(self._env, buttons)

    def step_frame(self, action_int, channels):
        """apply_ale_action(action) followed by the frame of the new state -- what ToyboxBaseEnv.step needs -- in ONE round trip
        to the engine (tbx_step1_frame).  Returns the (H, W, channels) uint8 frame; score and lives are cached for the
        get_score / get_lives / game_over calls that follow."""
        if int(action_int) not in self._engine.legal_actions:
            raise ValueError("Expected to apply action, but failed: {0}".format(action_int))
        # only the last sub-frame is rasterised (a frame per sub-frame is a render launch, a copy and a sync each on batch engines)
        for _ in range(self.frames_per_action - 1):
            self._engine.step1(self._env, action_int)
        _, _, lives, score, frame = self._engine.step1_frame(self._env, action_int, channels)
        self._scal = (score, lives)
        return frame

    def apply_action(self, action_input_obj):
        if not isinstance(action_input_obj, Input):
            raise TypeError("apply_action takes an Input")
        self._scal = None
        for _ in range(self.frames_per_action):
            self._engine.apply_input(self._env, action_input_obj.to_mask())

    # ------------------------------------------------------------------ T5: scalars
    def _scalar(self, which):
        if self._scal is not None and which != "level":
            score, lives = self._scal
            return {"score": score, "lives": lives, "over": lives <= 0}[which]
        score, lives, level, over = self._engine.scalars()
        return {"score": int(score[self._env]), "lives": int(lives[self._env]), "level": int(level[self._env]),
                "over": bool(over[self._env])}[which]

    def get_score(self):
        return self._scalar("score")

    def get_lives(self):
        return self._scalar("lives")

    def get_level(self):
        return self._scalar("level")

    def game_over(self):
        return self._scalar("over")

    # ------------------------------------------------------------------ R: frames
    def get_height(self):
        return self._engine.height

    def get_width(self):
        return self._engine.width

    def get_state(self):
        """Rendered frame: (H,W,1) gray when self.grayscale else (H,W,4) RGBA (envs/atari/base.py:108-113)."""
        return self._engine.render_env(self._env, 1 if self.grayscale else 4)

    def get_rgb_frame(self):
        return self._engine.render_env(self._env, 3)

    def save_frame_image(self, path, grayscale=False):
        if isinstance(path, bytes):
            path = path.decode("utf-8")
        write_png(path, self._engine.render_env(self._env, 1 if grayscale else 3))

    # ------------------------------------------------------------------ J: JSON state / config
    def state_to_json(self):
        return self._codec.state_to_json(self._engine.get_state(self._env))

    def to_state_json(self):
        return self.state_to_json()

    def write_state_json(self, js):
        self._scal = None
        if isinstance(js, (str, bytes)):
            js = json.loads(js)
        self._engine.set_state(self._env, self._codec.state_from_json(js))

    def config_to_json(self):
        cfg = self._engine.get_config()
        r = self._engine.get_sim_rng(self._env)
        cfg.rand[0], cfg.rand[1] = r
        return self._codec.config_to_json(cfg)

    def write_config_json(self, config_js):
        """Replaces the simulator config (batch-wide on a shared engine) and starts a new game."""
        if isinstance(config_js, (str, bytes)):
            config_js = json.loads(config_js)
        self._scal = None
        self._engine.set_config(self._codec.config_from_json(config_js))
        self.new_game()

    def schema_for_state(self):
        return self._codec.schema_for_state()

    def schema_for_config(self):
        return self._codec.schema_for_config()

    def query_state_json(self, query, args="null"):
        if isinstance(args, (str, bytes)):
            args = json.loads(args)
        return self._codec.query(self, query, args)


class Simulator(object):
    """Name-compatible view of the simulator half of a Toybox (config + RNG)."""

    def __init__(self, toybox_or_name):
        self._tb = toybox_or_name if isinstance(toybox_or_name, Toybox) else Toybox(toybox_or_name)
        self.game_name = self._tb.game_name

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def set_seed(self, seed):
        self._tb.set_seed(seed)

    def get_frame_width(self):
        return self._tb.get_width()

    def get_frame_height(self):
        return self._tb.get_height()

    def get_simulator(self):
        return self

    def new_game(self):
        self._tb.new_game()
        return State(self._tb)

    def to_json(self):
        return self._tb.config_to_json()

    def from_json(self, config_js):
        self._tb.write_config_json(config_js)

    def schema_for_state(self):
        return self._tb.schema_for_state()

    def schema_for_config(self):
        return self._tb.schema_for_config()


class State(object):
    """Name-compatible view of the state half of a Toybox."""

    def __init__(self, toybox):
        self._tb = toybox

    def __bool__(self):
        return True

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def lives(self):
        return self._tb.get_lives()

    def score(self):
        return self._tb.get_score()

    def level(self):
        return self._tb.get_level()

    def game_over(self):
        return self._tb.game_over()

    def query_json(self, query, args="null"):
        return self._tb.query_state_json(query, args)

    def render_frame(self, sim=None, grayscale=True):
        return self._tb._engine.render_env(self._tb._env, 1 if grayscale else 4)

    def render_frame_color(self, sim=None):
        return self._tb._engine.render_env(self._tb._env, 4)

    def render_frame_rgb(self, sim=None):
        return self._tb._engine.render_env(self._tb._env, 3)

    def render_frame_grayscale(self, sim=None):
        return self._tb._engine.render_env(self._tb._env, 1)

    def to_json(self):
        return self._tb.state_to_json()
