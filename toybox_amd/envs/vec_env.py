"""Batched env with the VecEnv contract of the reference's vendored baselines
(/root/reference/baselines/baselines/common/vec_env/__init__.py:26-131): reset() -> obs[N,...];
step_async(actions[N]); step_wait() -> (obs, rews float32, dones bool, infos); a done env is reset and
the returned obs is the reset obs (dummy_vec_env.py:51-54, subproc_vec_env.py:11-15).  The N worker
processes + pipes of SubprocVecEnv collapse into one device batch: one step kernel + one render kernel."""
import numpy as np

from .. import _abi
from ..engine import Engine
from ..games import codec
from ..toybox import _make_engine
from .base import hash_seed
from .spaces import Box, Discrete


class LazyInfos:
    """The per-env info dicts of a VecEnv step as a read-only sequence that builds a dict only when one is asked for:
    65 536 Python dicts per step would cost more host time than the whole device step.  Indexing, iteration and len()
    behave like the list of dicts baselines expects (`for info in infos: info.get('episode')`)."""

    def __init__(self, n, columns=None, extras=None):
        self._n = int(n)
        self._columns = columns or {}        # key -> array[N]
        self._extras = extras or {}          # env index -> dict of additional keys

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        d = {k: v[i].item() for k, v in self._columns.items()}
        d.update(self._extras.get(i, ()))
        return d

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def with_key(self, key):
        """{env index: value} of the envs whose info carries `key` -- e.g. infos.with_key('episode') -- without walking N dicts"""
        if key in self._columns:
            return {i: self._columns[key][i].item() for i in range(self._n)}
        return {i: d[key] for i, d in self._extras.items() if key in d}


class ToyboxVecEnv:
    def __init__(self, game, num_envs, grayscale=True, alpha=False, seed=None, cache_terminal_state=False, engine=None,
                 reuse_obs_buffer=False):
        """reuse_obs_buffer: observations are written into ONE page-locked host array that every reset() / step() returns
        again (the copy then runs at the PCIe link's rate; a fresh pageable array per call, the default and what the
        reference's VecEnvs hand out, costs a staged copy and page faults -- bench.py --protocol host times both)."""
        self.game = {"spaceinvaders": "space_invaders"}.get(game, game)
        self.num_envs = int(num_envs)
        self.engine = engine if engine is not None else _make_engine(self.game, self.num_envs)
        self._channels = 1 if grayscale else 4 if alpha else 3
        self._obs_buf = None
        if reuse_obs_buffer:
            from ..hip import PinnedArray
            self._obs_buf = PinnedArray((self.num_envs, self.engine.height, self.engine.width, self._channels))
        self._action_set = sorted(self.engine.legal_actions)
        self._lut = np.asarray(self._action_set, dtype=np.int32)
        self.action_space = Discrete(len(self._action_set))
        self.observation_space = Box(0, 255, (self.engine.height, self.engine.width, self._channels), "uint8")
        self.cache_terminal_state = bool(cache_terminal_state)
        self._codec = codec(self.game)
        self._pending = None
        self.closed = False
        if seed is not None:
            self.seed(seed)

    # ------------------------------------------------------------------ seeding
    def seed(self, seed):
        """env i gets the reference's per-env derivation (cmd_util.py:31: seed + rank), each pushed through
        ToyboxBaseEnv.seed's hash (envs/atari/base.py:84-98); returns [[seed1, seed2], ...]."""
        out = [[int(seed) + i, hash_seed(int(seed) + i + 1) % 2 ** 31] for i in range(self.num_envs)]
        self.engine.seed_array([s2 for _, s2 in out])      # one upload + one launch, not N
        self.engine.new_game()
        return out

    # ------------------------------------------------------------------ VecEnv
    def _frames(self):
        return self.engine.render(self._channels, out=self._obs_buf.array if self._obs_buf is not None else None)

    def reset(self):
        self.engine.new_game()
        return self._frames()

    def step_async(self, actions):
        a = np.asarray(actions)
        if a.shape != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        if a.min() < 0 or a.max() >= len(self._action_set):
            raise AssertionError("action index out of range")   # assert action_index < len(action_set) (base.py:123)
        self._pending = self._lut[a.astype(np.int64)]

    def step_wait(self):
        assert self._pending is not None, "step_wait without step_async"
        actions, self._pending = self._pending, None
        extras = {}
        if self.cache_terminal_state:
            reward, done, lives, score = self.engine.step(actions, auto_reset=False)
            idx = np.nonzero(done)[0]
            for i in idx:
                extras[int(i)] = {"cached_state": self._codec.state_to_json(self.engine.get_state(int(i)))}
            if len(idx):
                self.engine.new_game(done.astype(np.uint8))
        else:
            reward, done, lives, score = self.engine.step(actions, auto_reset=True)
        obs = self._frames()
        infos = LazyInfos(self.num_envs, {"lives": lives, "score": np.where(done, 0, score)}, extras)
        return obs, reward.astype(np.float32), done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        return self.engine.render(3)

    def render(self, mode="rgb_array"):
        imgs = self.get_images()
        if mode == "rgb_array":
            return imgs
        raise NotImplementedError("only rgb_array rendering is provided")

    def close(self):
        if not self.closed:
            self.engine.close()
            if self._obs_buf is not None:
                self._obs_buf.close()
            self.closed = True

    @property
    def unwrapped(self):
        return self

    def get_action_meanings(self):
        from .constants import ACTION_MEANING
        return list(ACTION_MEANING.values())


class ToyboxPreprocVecEnv:
    """The DeepMind-style pipeline of the reference's baselines fork -- MaxAndSkipEnv(skip) -> WarpFrame(84x84 gray) ->
    ClipRewardEnv -> VecFrameStack(stack) (atari_wrappers.py:193-244,324-360; vec_frame_stack.py:17-30) -- as ONE device
    pass per agent step (tbx_agent_step): the learner only ever sees uint8[N, size, size, stack].

    episode_life / fire_reset / noop_max switch on the reset-time wrappers of wrap_deepmind / make_atari
    (EpisodicLifeEnv :58-96, FireResetEnv :38-56, NoopResetEnv :12-36), run inside the reset kernel; the Monitor
    record of a finished game arrives as info["episode"] = {"r", "l"} like bench/monitor.py:68-76.  env_offset is the
    global index of env 0 (multi-GPU sharding) so no-op counts do not depend on the shard layout.

    frame_stack="vec" (default) stacks like VecFrameStack over the vector env: a reset leaves zeros in the older slots.
    frame_stack="env" stacks like wrap_deepmind(frame_stack=True), a FrameStack(k) inside every env
    (atari_wrappers.py:246-275): a reset fills the whole stack with its observation.  scale=True is ScaledFloatFrame
    (atari_wrappers.py:277-286): observations come back as float32 in [0, 1] (uint8 / 255, converted on the host; the device
    buffer TBX_BUF_AGENT_OBS stays uint8)."""

    def __init__(self, game, num_envs, skip=4, size=84, stack=4, clip_rewards=True, seed=None, engine=None,
                 episode_life=False, fire_reset=False, noop_max=0, noop_seed=0, env_offset=0, frame_stack="vec", scale=False,
                 reuse_obs_buffer=False):
        self.game = {"spaceinvaders": "space_invaders"}.get(game, game)
        self.num_envs = int(num_envs)
        self.engine = engine if engine is not None else _make_engine(self.game, self.num_envs)
        self._action_set = sorted(self.engine.legal_actions)
        self._lut = np.asarray(self._action_set, dtype=np.int32)
        self.action_space = Discrete(len(self._action_set))
        if frame_stack not in ("vec", "env"):
            raise ValueError("frame_stack must be 'vec' (VecFrameStack) or 'env' (FrameStack inside every env)")
        self.scale = bool(scale)
        self.observation_space = Box(0, 1.0, (size, size, stack), "float32") if self.scale else Box(0, 255, (size, size, stack), "uint8")
        if seed is not None:
            self.engine.seed_array([hash_seed(int(seed) + i + 1) % 2 ** 31 for i in range(self.num_envs)])
        self.engine.agent_init(skip=skip, out_h=size, out_w=size, stack=stack, clip_reward=clip_rewards,
                               episodic_life=episode_life, fire_reset=fire_reset, noop_max=noop_max, noop_seed=noop_seed,
                               env_offset=env_offset, stack_fill=1 if frame_stack == "env" else 0)
        self._pending = None
        self.closed = False
        self._obs_buf = None                                 # reuse_obs_buffer: as in ToyboxVecEnv (the uint8 observations)
        if reuse_obs_buffer:
            from ..hip import PinnedArray
            self._obs_buf = PinnedArray((self.num_envs, size, size, stack))

    def _out(self):
        return self._obs_buf.array if self._obs_buf is not None else None

    def _obs(self, obs):
        # ScaledFloatFrame.observation: np.array(observation).astype(np.float32) / 255.0
        return obs.astype(np.float32) / 255.0 if self.scale else obs

    def reset(self):
        return self._obs(self.engine.agent_reset(out=self._out()))

    def step_async(self, actions):
        a = np.asarray(actions)
        if a.shape != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        if a.min() < 0 or a.max() >= len(self._action_set):
            raise AssertionError("action index out of range")
        self._pending = self._lut[a.astype(np.int64)]

    def step_wait(self):
        assert self._pending is not None, "step_wait without step_async"
        actions, self._pending = self._pending, None
        obs, reward, done = self.engine.agent_step(actions, out=self._out())
        ended, ret, length = self.engine.agent_episodes()
        extras = {int(i): {"episode": {"r": float(ret[i]), "l": int(length[i])}} for i in np.flatnonzero(ended)}
        return self._obs(obs), reward, done, LazyInfos(self.num_envs, None, extras)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if not self.closed:
            self.engine.close()
            if self._obs_buf is not None:
                self._obs_buf.close()
            self.closed = True
