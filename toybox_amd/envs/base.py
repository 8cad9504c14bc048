"""Single-env gym-style surface with the semantics of the reference's ToyboxBaseEnv
(/root/reference/toybox/envs/atari/base.py:38-173): obs = rendered frame (H,W,C) uint8 with
C = 1 if grayscale else 4 if alpha else 3; reward = max(score - previous score, 0); done = lives <= 0;
info = {lives, score (0 when done), cached_state on the game-over step}.  gym itself is optional."""
import hashlib

import numpy as np

from ..toybox import Toybox
from .constants import ACTION_MEANING
from .spaces import Box, Discrete


def hash_seed(seed, max_bytes=8):
    """gym.utils.seeding.hash_seed of the gym era the reference targets: little-endian int of sha512(str(seed))[:8]."""
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return int.from_bytes(h[:max_bytes], "little")


class MockALE:
    """ALE-shaped view over a Toybox (envs/atari/base.py:15-35)."""

    def __init__(self, toybox):
        self.toybox = toybox

    def lives(self):
        return self.toybox.get_lives()

    def get_score(self):
        return self.toybox.get_score()

    def game_over(self):
        # matches baselines / atari_py, not what videogames would expect (envs/atari/base.py:25-27)
        return self.toybox.get_lives() <= 0

    def saveScreenPNG(self, name):
        if isinstance(name, bytes):
            name = name.decode("utf-8")
        self.toybox.save_frame_image(name, grayscale=False)


class ToyboxBaseEnv:
    metadata = {"render.modes": ["human", "rgb_array"]}
    reward_range = (0, float("inf"))
    game_name = None

    def __init__(self, toybox=None, game=None, frameskip=(2, 5), repeat_action_probability=0.0, grayscale=True,
                 alpha=False, actions=None):
        if toybox is None:
            toybox = Toybox(game or self.game_name, grayscale)
        assert toybox.rstate
        self.toybox = toybox
        self.cached_state = None
        self.score = self.toybox.get_score()
        self.viewer = None
        self._np_random = None
        self.ale = MockALE(toybox)
        if actions is None:
            actions = toybox.get_legal_action_set()
        assert actions is not None
        self._action_set = list(actions)
        self._obs_type = "image"
        self._rgba = 1 if grayscale else 4 if alpha else 3
        self._pixel_high = 255
        self._height = self.toybox.get_height()
        self._width = self.toybox.get_width()
        self._dim = (self._height, self._width, self._rgba)
        self.action_space = Discrete(len(self._action_set))
        self.observation_space = Box(low=0, high=self._pixel_high, shape=self._dim, dtype="uint8")

    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed=None):
        """envs/atari/base.py:84-98: seed1 -> seed2 = hash_seed(seed1 + 1) % 2**31 -> set_seed -> new_game."""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy % (2 ** 31))
        seed1 = int(seed)
        self._np_random = np.random.RandomState(seed1 % (2 ** 32))
        seed2 = hash_seed(seed1 + 1) % 2 ** 31
        self.toybox.set_seed(seed2)
        self.toybox.new_game()
        return [seed1, seed2]

    def get_action_meanings(self):
        # all 18 names regardless of the action set, as the reference does (envs/atari/base.py:102-104)
        return list(ACTION_MEANING.values())

    def _get_obs(self):
        if self._rgba == 1:
            return self.toybox._engine.render_env(self.toybox._env, 1)
        return self.toybox._engine.render_env(self.toybox._env, self._rgba)

    def step(self, action_index):
        info = {}
        assert action_index < len(self._action_set)
        self.toybox.apply_ale_action(self._action_set[int(action_index)])
        if self.ale.game_over():
            info["cached_state"] = self.toybox.to_state_json()
        obs = self._get_obs()
        score = self.toybox.get_score()
        reward = max(score - self.score, 0)
        self.score = score
        done = self.ale.game_over()
        info["lives"] = self.toybox.get_lives()
        info["score"] = 0 if done else self.score
        return obs, reward, done, info

    def reset(self):
        self.cached_state = self.toybox.to_state_json()
        self.toybox.new_game()
        self.score = self.toybox.get_score()
        return self._get_obs()

    def render(self, mode="human", close=False):
        if mode == "rgb_array":
            return self.toybox.get_rgb_frame()
        if mode == "human":
            return self.toybox.get_rgb_frame()   # no viewer dependency: hand the frame to the caller
        raise ValueError("unknown render mode %r" % (mode,))

    def close(self):
        if self.toybox is not None:
            self.toybox.close()
        self.toybox = None


class BreakoutEnv(ToyboxBaseEnv):
    game_name = "breakout"

    def __init__(self, frameskip=(2, 5), repeat_action_probability=0.0, grayscale=True, alpha=False):
        super().__init__(Toybox("breakout", grayscale), "breakout", frameskip, repeat_action_probability,
                         grayscale=grayscale, alpha=alpha)


class AmidarEnv(ToyboxBaseEnv):
    game_name = "amidar"

    def __init__(self, frameskip=(2, 5), repeat_action_probability=0.0, grayscale=True, alpha=False):
        super().__init__(Toybox("amidar", grayscale), "amidar", frameskip, repeat_action_probability,
                         grayscale=grayscale, alpha=alpha)


class SpaceInvadersEnv(ToyboxBaseEnv):
    game_name = "space_invaders"

    def __init__(self, frameskip=(2, 5), repeat_action_probability=0.0, grayscale=True, alpha=False):
        super().__init__(Toybox("space_invaders", grayscale), "space_invaders", frameskip, repeat_action_probability,
                         grayscale=grayscale, alpha=alpha)


class GridWorldEnv(ToyboxBaseEnv):
    """envs/atari/gridworld.py:8-13 (frameskip (0, 0) there; neither exported nor gym-registered by the reference)."""
    game_name = "gridworld"

    def __init__(self, frameskip=(0, 0), repeat_action_probability=0.0, grayscale=True, alpha=False):
        super().__init__(Toybox("gridworld", grayscale), "gridworld", frameskip, repeat_action_probability,
                         grayscale=grayscale, alpha=alpha)


ENV_IDS = {
    # gym ids of the reference (toybox/__init__.py:8-24)
    "BreakoutToyboxNoFrameskip-v4": BreakoutEnv,
    "AmidarToyboxNoFrameskip-v4": AmidarEnv,
    "SpaceInvadersToyboxNoFrameskip-v4": SpaceInvadersEnv,
}


def make(env_id, **kwargs):
    return ENV_IDS[env_id](**kwargs)


def register_with_gym():
    """Registers the three ids with gym when gym is installed (toybox/__init__.py:8-24)."""
    from gym.envs.registration import register
    register(id="BreakoutToyboxNoFrameskip-v4", entry_point="toybox_amd.envs:BreakoutEnv", nondeterministic=True)
    register(id="AmidarToyboxNoFrameskip-v4", entry_point="toybox_amd.envs:AmidarEnv", nondeterministic=False)
    register(id="SpaceInvadersToyboxNoFrameskip-v4", entry_point="toybox_amd.envs:SpaceInvadersEnv", nondeterministic=False)
