"""One-env gym surface over an engine: the drop-in for the reference's `toybox.envs.atari` classes
(/root/reference/toybox/envs/atari/base.py:38-173, breakout.py / amidar.py / space_invaders.py / gridworld.py).

Contract kept from the reference (each line checked by tests/test_envs.py on both libraries, and -- through the reference's own
wrapper classes -- by the fixtures of tests/golden/wrappers/):
  step(i)  -> (frame, reward, done, info): i indexes the sorted legal action set; frame is (H, W, C) uint8 with C = 1 for
              grayscale, 4 with alpha, else 3; reward = max(score - score at the previous step, 0); done = lives <= 0;
              info = {"lives", "score" (0 once done), "cached_state" (state JSON, on the game-over step only)}
  reset()  -> frame of a new game; the state before it is kept in `cached_state`
  seed(s)  -> [s, hash_seed(s + 1) % 2**31]; the second value seeds the simulator and a new game is started
  .ale     -> lives() / get_score() / game_over() / saveScreenPNG(name), what baselines' wrappers ask an ALE for
  .toybox  -> the ctoybox-shaped Toybox (interventions take it from here)
The class derives from gym.Env when a `gym` is importable and registers the reference's three ids with it (envs/__init__.py);
without gym it is a plain class with the same methods and `toybox_amd.envs.make(id)`.
"""
import hashlib
import os

import numpy as np

from ..toybox import Toybox
from .constants import ACTION_MEANING

try:                                                    # gym is optional (absent from the ROCm image)
    import gym as _gym
    from gym.spaces import Box, Discrete
    _EnvBase = _gym.Env
except ImportError:                                     # pragma: no cover - depends on the installation
    _gym = None
    from .spaces import Box, Discrete
    _EnvBase = object


def hash_seed(seed, max_bytes=8):
    """gym.utils.seeding.hash_seed of the gym era the reference targets: little-endian int of sha512(str(seed))[:8]."""
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()
    return int.from_bytes(digest[:max_bytes], "little")


def _int_list_from_bigint(bigint):
    """gym.utils.seeding._int_list_from_bigint: little-endian 32-bit words, none for the zero high part ([0] for 0)"""
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


class MockALE:
    """What baselines' wrappers read from `env.unwrapped.ale` (EpisodicLifeEnv: lives(); envs/atari/base.py:15-35)."""

    def __init__(self, toybox):
        self.toybox = toybox

    def lives(self):
        return self.toybox.get_lives()

    def get_score(self):
        return self.toybox.get_score()

    def game_over(self):
        return self.lives() <= 0                        # "out of lives", the atari_py meaning, not the game's own flag

    def saveScreenPNG(self, name):
        path = name.decode("utf-8") if isinstance(name, bytes) else name
        self.toybox.save_frame_image(path, grayscale=False)


class ToyboxBaseEnv(_EnvBase):
    metadata = {"render.modes": ["human", "rgb_array"]}
    reward_range = (0, float("inf"))
    game_name = None                                    # set by the per-game subclasses

    def __init__(self, toybox=None, game=None, frameskip=(2, 5), repeat_action_probability=0.0, grayscale=True, alpha=False,
                 actions=None):
        # frameskip / repeat_action_probability: accepted for signature compatibility; like the reference's NoFrameskip
        # envs, one step is one frame and actions are never repeated at random
        tb = toybox if toybox is not None else Toybox(game or self.game_name, grayscale)
        if not tb.rstate:
            raise ValueError("the Toybox handed to %s has been closed" % type(self).__name__)
        self.toybox = tb
        self.ale = MockALE(tb)
        self.channels = 1 if grayscale else (4 if alpha else 3)
        self._action_set = list(tb.get_legal_action_set() if actions is None else actions)
        self.action_space = Discrete(len(self._action_set))
        self.observation_space = Box(low=0, high=255, shape=(tb.get_height(), tb.get_width(), self.channels), dtype=np.uint8)
        self.score = tb.get_score()                     # the score one step ago: rewards are its increases
        self.cached_state = None
        self._np_random = None
        self.viewer = None                              # (headless: never created; the attribute is part of the reference's surface)
        # the private names the reference's class carries (envs/atari/base.py:60-66), for code that peeks at them
        self._rgba = self.channels
        self._height, self._width = tb.get_height(), tb.get_width()
        self._dim = (self._height, self._width, self._rgba)
        self._obs_type, self._pixel_high = "image", 255

    # ------------------------------------------------------------------ gym.Env
    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        """NoopResetEnv draws its no-op count here (atari_wrappers.py:124); seeded lazily like gym's envs"""
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed=None):
        # gym.utils.seeding.np_random of the gym era the reference targets (envs/atari/base.py:84-98), restated rather than
        # called: from gym 0.26 on that function returns a numpy Generator, which has no .randint (NoopResetEnv calls
        # unwrapped.np_random.randint, atari_wrappers.py:124) and draws another stream.  create_seed: an int is taken modulo
        # 2**64, None becomes 8 bytes of OS entropy; the RandomState is seeded with the little-endian 32-bit words of
        # hash_seed(seed) -- zero high words dropped, as _int_list_from_bigint leaves them out -- not with the seed itself.
        if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
            raise ValueError("Seed must be a non-negative integer or omitted, not %r" % (seed,))
        first = int.from_bytes(os.urandom(8), "big") if seed is None else int(seed) % 2 ** 64
        self._np_random = np.random.RandomState()
        self._np_random.seed(_int_list_from_bigint(hash_seed(first)))
        second = hash_seed(first + 1) % 2 ** 31
        self.toybox.set_seed(second)
        self.toybox.new_game()                          # the simulator's seed only acts through a new game
        return [first, second]

    def get_action_meanings(self):
        return list(ACTION_MEANING.values())            # all 18 ALE names whatever the action set, as the reference answers

    def _frame(self):
        return self.toybox._engine.render_env(self.toybox._env, self.channels)

    def _get_obs(self):
        """the reference's name for it (envs/atari/base.py:106-113): the current frame, gray (H, W, 1) or colour without / with alpha"""
        return self._frame()

    def step(self, action_index):
        if not 0 <= action_index < len(self._action_set):
            raise AssertionError("action index %r outside the %d legal actions" % (action_index, len(self._action_set)))
        tb = self.toybox
        # the frame's step and its picture in one round trip to the engine (a one-env engine answers both from its resident
        # kernel: no launch, no copy, no synchronisation)
        frame = tb.step_frame(self._action_set[int(action_index)], self.channels)
        score, lives = tb.get_score(), tb.get_lives()
        done = lives <= 0
        info = {"lives": lives, "score": 0 if done else score}
        if done:
            info["cached_state"] = tb.to_state_json()
        reward, self.score = max(score - self.score, 0), score
        return frame, reward, done, info

    def reset(self):
        tb = self.toybox
        self.cached_state = tb.to_state_json()
        tb.new_game()
        self.score = tb.get_score()
        return self._frame()

    def render(self, mode="human", close=False):
        if mode not in ("human", "rgb_array"):
            raise ValueError("unknown render mode %r" % (mode,))
        return self.toybox.get_rgb_frame()              # headless: "human" hands the frame to the caller, no viewer window

    def close(self):
        if self.toybox is not None:
            self.toybox.close()
        self.toybox = None


def _game_env(game, default_frameskip=(2, 5)):
    """the per-game classes differ only in the game they open (envs/atari/breakout.py:7-12 and siblings)"""

    class GameEnv(ToyboxBaseEnv):
        game_name = game

        def __init__(self, frameskip=default_frameskip, repeat_action_probability=0.0, grayscale=True, alpha=False):
            ToyboxBaseEnv.__init__(self, None, game, frameskip, repeat_action_probability, grayscale=grayscale, alpha=alpha)

    return GameEnv


BreakoutEnv = _game_env("breakout")
AmidarEnv = _game_env("amidar")
SpaceInvadersEnv = _game_env("space_invaders")
GridWorldEnv = _game_env("gridworld", (0, 0))           # envs/atari/gridworld.py:8-13: neither exported nor registered upstream
for _name, _cls in (("BreakoutEnv", BreakoutEnv), ("AmidarEnv", AmidarEnv), ("SpaceInvadersEnv", SpaceInvadersEnv),
                    ("GridWorldEnv", GridWorldEnv)):
    _cls.__name__ = _cls.__qualname__ = _name

# gym ids of the reference (toybox/__init__.py:8-24) and their `nondeterministic` flags
ENV_IDS = {
    "BreakoutToyboxNoFrameskip-v4": BreakoutEnv,
    "AmidarToyboxNoFrameskip-v4": AmidarEnv,
    "SpaceInvadersToyboxNoFrameskip-v4": SpaceInvadersEnv,
}
_NONDETERMINISTIC = {"BreakoutToyboxNoFrameskip-v4": True}


def make(env_id, **kwargs):
    return ENV_IDS[env_id](**kwargs)


def register_with_gym():
    """Puts the three ids into gym's registry (what `import toybox` does upstream); returns the ids it added."""
    from gym.envs.registration import register
    added = []
    for env_id, cls in ENV_IDS.items():
        try:
            register(id=env_id, entry_point="toybox_amd.envs:%s" % cls.__name__,
                     nondeterministic=_NONDETERMINISTIC.get(env_id, False))
            added.append(env_id)
        except Exception as exc:                        # gym raises on a second registration of the same id
            if "Cannot re-register" not in str(exc) and "already" not in str(exc).lower():
                raise
    return added
