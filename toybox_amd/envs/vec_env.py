"""Batched env with the VecEnv contract of the reference's vendored baselines
(/root/reference/baselines/baselines/common/vec_env/__init__.py:26-131): reset() -> obs[N,...];
step_async(actions[N]); step_wait() -> (obs, rews float32, dones bool, infos); a done env is reset and
the returned obs is the reset obs (dummy_vec_env.py:51-54, subproc_vec_env.py:11-15).  The N worker
processes + pipes of SubprocVecEnv collapse into one device batch: one step kernel + one render kernel."""
import time

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


def _pool_size(obs_pool, reuse_obs_buffer):
    """obs_pool with the older reuse_obs_buffer spelling folded in (True: one array, False: a fresh array per call)"""
    if reuse_obs_buffer is not None:
        return 1 if reuse_obs_buffer else 0
    p = int(obs_pool)
    if p < 0:
        raise ValueError("obs_pool must be >= 0")
    return p


class ToyboxVecEnv:
    CACHE_TERMINAL_STATE_UP_TO = 64      # cache_terminal_state=None: on up to this many envs

    def __init__(self, game, num_envs, grayscale=True, alpha=False, seed=None, cache_terminal_state=None, engine=None,
                 obs_pool=2, reuse_obs_buffer=None):
        """cache_terminal_state: ToyboxBaseEnv.step always attaches info["cached_state"] = the state JSON on the step that ends
        a game (envs/atari/base.py:128-130).  Here that needs the finished envs' state records on the host BEFORE they are reset,
        which takes the step out of the asynchronous path (step, read-back, new_game, render as four synchronous calls).  None
        (default): on for batches of up to CACHE_TERMINAL_STATE_UP_TO envs -- the sizes the reference's own vector envs run
        (run.py: nenv = the CPU count) -- and off above, where the batch step with in-kernel auto-reset is the point; True /
        False force it.  INTEGRATION.md section 3, row V.
        obs_pool: the observations that reset() / step() return are page-locked host arrays handed out in rotation, so an
        observation stays valid until obs_pool - 1 further steps have been taken (default 2: the reference's learners copy on
        receipt -- `self.obs[:] = self.env.step(...)`, baselines/ppo2/ppo2.py:110 -- and the copy from the device runs at the PCIe
        link's rate, asynchronously between step_async and step_wait).  obs_pool = 0 is the contract as the reference's VecEnvs
        spell it, a fresh pageable array per call: a staged copy plus page faults, several times slower
        (bench.py --protocol host times both).  reuse_obs_buffer=True/False is the older spelling of obs_pool = 1 / 0."""
        self.game = {"spaceinvaders": "space_invaders"}.get(game, game)
        self.num_envs = int(num_envs)
        self.engine = engine if engine is not None else _make_engine(self.game, self.num_envs)
        self._channels = 1 if grayscale else 4 if alpha else 3
        shape = (self.num_envs, self.engine.height, self.engine.width, self._channels)
        self._obs_shape = shape
        self._pool = [self.engine.host_array(shape) for _ in range(_pool_size(obs_pool, reuse_obs_buffer))]
        self._turn = 0
        # step outputs staged in page-locked memory; what step_wait returns are fresh copies (rollout buffers keep references:
        # `mb_rewards.append(rewards)`, ppo2.py:113)
        n = self.num_envs
        self._st = {"reward": self.engine.host_array((n,), np.int32), "done": self.engine.host_array((n,), np.uint8),
                    "lives": self.engine.host_array((n,), np.int32), "score": self.engine.host_array((n,), np.int32)}
        self._action_set = sorted(self.engine.legal_actions)
        self._lut = np.asarray(self._action_set, dtype=np.int32)
        self.action_space = Discrete(len(self._action_set))
        self.observation_space = Box(0, 255, (self.engine.height, self.engine.width, self._channels), "uint8")
        self.cache_terminal_state = self.num_envs <= self.CACHE_TERMINAL_STATE_UP_TO if cache_terminal_state is None else bool(cache_terminal_state)
        self._codec = codec(self.game)
        self._pending = None
        self._in_flight = None
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
    def _next_obs_array(self):
        if not self._pool:
            return np.empty(self._obs_shape, np.uint8)
        a = self._pool[self._turn % len(self._pool)]
        self._turn += 1
        return a

    def _frames(self):
        return self.engine.render(self._channels, out=self._next_obs_array())

    def reset(self):
        if self._in_flight is not None:          # a step between step_async and step_wait: it ends first (its results are dropped)
            self.step_wait()
        self._pending = None
        self.engine.new_game()
        return self._frames()

    def step_async(self, actions):
        """Queues the whole step -- actions to the device, the batch step with auto-reset, the rasteriser, the copy of every
        frame and of the step outputs to host memory -- and returns; step_wait() collects
        (vec_env/__init__.py:67-87, subproc_vec_env.py:63-74)."""
        a = np.asarray(actions)
        if a.shape != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        if a.min() < 0 or a.max() >= len(self._action_set):
            raise AssertionError("action index out of range")   # assert action_index < len(action_set) (base.py:123)
        ale = self._lut[a.astype(np.int64)]
        if self.cache_terminal_state:
            self._pending = ale                                  # (needs the states of finished games before they are reset)
            return
        obs = self._next_obs_array()
        st = self._st
        self.engine.step_begin(ale, auto_reset=True, reward=st["reward"], done=st["done"], lives=st["lives"], score=st["score"],
                               frame=obs, channels=self._channels)
        self._in_flight = obs

    def step_wait(self):
        if self._in_flight is not None:
            obs, self._in_flight = self._in_flight, None
            self.engine.step_end()
            st = self._st
            reward, lives, score = st["reward"].copy(), st["lives"].copy(), st["score"].copy()
            done = st["done"].astype(bool)
            infos = LazyInfos(self.num_envs, {"lives": lives, "score": np.where(done, 0, score)}, {})
            return obs, reward.astype(np.float32), done, infos
        assert self._pending is not None, "step_wait without step_async"
        actions, self._pending = self._pending, None
        extras = {}
        reward, done, lives, score = self.engine.step(actions, auto_reset=False)
        idx = np.nonzero(done)[0]
        for i in idx:
            extras[int(i)] = {"cached_state": self._codec.state_to_json(self.engine.get_state(int(i)))}
        if len(idx):
            self.engine.new_game(done.astype(np.uint8))
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
            if self._in_flight is not None:
                self.step_wait()
            self.engine.close()                 # (the page-locked arrays live on while a caller holds an observation)
            self.closed = True

    @property
    def unwrapped(self):
        return self

    def get_action_meanings(self):
        from .constants import ACTION_MEANING
        return list(ACTION_MEANING.values())


class PlaneStack:
    """The stacked observation of every env as its `stack` planes, oldest first, each uint8[N, h, w] in page-locked host memory
    -- the batched counterpart of the reference's LazyFrames (atari_wrappers.py:288-317: "common frames between the observations
    are only stored once ... should only be converted to numpy array before being passed to the model").  np.asarray(obs) is
    the uint8[N, h, w, stack] array VecFrameStack / FrameStack would have returned; obs[i] one env's (h, w, stack); obs.planes
    the planes themselves (no copy).  Valid until the env's next step: the planes are shared with the following observations
    and VecFrameStack's zeroing of a finished env's older frames (vec_frame_stack.py:21-23) happens in place."""

    def __init__(self, planes):
        self.planes = tuple(planes)
        n, h, w = self.planes[0].shape
        self.shape = (n, h, w, len(self.planes))
        self.dtype = np.dtype(np.uint8)
        self.ndim = 4

    def __array__(self, dtype=None, copy=None):
        out = np.stack(self.planes, axis=-1)
        return out if dtype is None else out.astype(dtype)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            return np.stack([p[i] for p in self.planes], axis=-1)
        return np.asarray(self)[i]

    def astype(self, dtype):
        return np.asarray(self).astype(dtype)


class ToyboxPreprocVecEnv:
    """The DeepMind-style pipeline of the reference's baselines fork -- MaxAndSkipEnv(skip) -> WarpFrame(84x84 gray) ->
    ClipRewardEnv -> VecFrameStack(stack) (atari_wrappers.py:193-244,324-360; vec_frame_stack.py:17-30) -- as ONE device
    pass per agent step (tbx_agent_step): the learner only ever sees uint8[N, size, size, stack].

    episode_life / fire_reset / noop_max switch on the reset-time wrappers of wrap_deepmind / make_atari
    (EpisodicLifeEnv :58-96, FireResetEnv :38-56, NoopResetEnv :12-36), run inside the reset kernel; the Monitor
    record of a finished game arrives as info["episode"] = {"r", "l", "t"} like bench/monitor.py:64-76 (t = seconds since this
    adapter was constructed, where Monitor / VecMonitor count from their own construction; 6 decimals).  env_offset is the
    global index of env 0 (multi-GPU sharding) so no-op counts do not depend on the shard layout.

    frame_stack="vec" (default) stacks like VecFrameStack over the vector env: a reset leaves zeros in the older slots.
    frame_stack="env" stacks like wrap_deepmind(frame_stack=True), a FrameStack(k) inside every env
    (atari_wrappers.py:246-275): a reset fills the whole stack with its observation.  scale=True is ScaledFloatFrame
    (atari_wrappers.py:277-286): observations come back as float32 in [0, 1] (uint8 / 255, converted on the host; the device
    buffer TBX_BUF_AGENT_OBS stays uint8).

    How the observation reaches the host (obs_layout; results are the same array values in every case):
      "device_stack" (default)  the stacks are rolled on the device and the whole uint8[N, size, size, stack] crosses PCIe into a
                                rotating pool of obs_pool page-locked arrays (see ToyboxVecEnv);
      "planes"                  the reference's data flow: ONE new size x size plane per env and step crosses PCIe
                                (subproc_vec_env.py:63-74 sends one frame per env; VecFrameStack stacks on the receiving side,
                                vec_frame_stack.py:17-30) into a ring of page-locked planes, and the observation is a
                                PlaneStack over the newest `stack` of them -- a quarter of the bytes, no roll at all (nor on
                                the device: tbx_agent_config_t::new_plane = 2, the observation kernels write the plane alone); a
                                finished env's older planes are zeroed (frame_stack="env": overwritten with the reset
                                observation) in place, exactly the values VecFrameStack / FrameStack produce;
      "host_stack"              the same one-plane transfer, then VecFrameStack's roll on the host into a real
                                uint8[N, size, size, stack] array of the rotating pool (tbx_host_stack_push: np.roll's data
                                movement as threaded host code; 9 bytes of host memory traffic per pixel, so at 10^4 envs it is
                                the host's memory system that sets the rate, not the link).
    step_async() queues the device work and the copies; step_wait() waits for them (vec_env/__init__.py:67-87)."""

    def __init__(self, game, num_envs, skip=4, size=84, stack=4, clip_rewards=True, seed=None, engine=None,
                 episode_life=False, fire_reset=False, noop_max=0, noop_seed=0, env_offset=0, frame_stack="vec", scale=False,
                 obs_pool=2, obs_layout="device_stack", reuse_obs_buffer=None):
        self.game = {"spaceinvaders": "space_invaders"}.get(game, game)
        self.num_envs = int(num_envs)
        self.engine = engine if engine is not None else _make_engine(self.game, self.num_envs)
        self._action_set = sorted(self.engine.legal_actions)
        self._lut = np.asarray(self._action_set, dtype=np.int32)
        self.action_space = Discrete(len(self._action_set))
        if frame_stack not in ("vec", "env"):
            raise ValueError("frame_stack must be 'vec' (VecFrameStack) or 'env' (FrameStack inside every env)")
        if obs_layout not in ("device_stack", "planes", "host_stack"):
            raise ValueError("obs_layout must be 'device_stack', 'planes' or 'host_stack'")
        self.obs_layout = obs_layout
        self._fill_repeat = frame_stack == "env"
        self.scale = bool(scale)
        self.stack, self.size = int(stack), int(size)
        self.observation_space = Box(0, 1.0, (size, size, stack), "float32") if self.scale else Box(0, 255, (size, size, stack), "uint8")
        if seed is not None:
            self.engine.seed_array([hash_seed(int(seed) + i + 1) % 2 ** 31 for i in range(self.num_envs)])
        self.engine.agent_init(skip=skip, out_h=size, out_w=size, stack=stack, clip_reward=clip_rewards,
                               episodic_life=episode_life, fire_reset=fire_reset, noop_max=noop_max, noop_seed=noop_seed,
                               env_offset=env_offset, stack_fill=1 if frame_stack == "env" else 0,
                               new_plane=0 if obs_layout == "device_stack" else 2)     # host layouts: no stack on the device at all
        self._pending = None
        self._in_flight = None
        self.closed = False
        self.tstart = time.time()                # Monitor.tstart (bench/monitor.py:19), VecMonitor.tstart (vec_monitor.py:12)
        n = self.num_envs
        pool = _pool_size(obs_pool, reuse_obs_buffer)
        self._pool, self._turn = [], 0
        if obs_layout != "planes":
            self._pool = [self.engine.host_array((n, size, size, stack)) for _ in range(pool)]
        if obs_layout != "device_stack":
            # the plane ring: the `stack` planes of the current observation + one per further observation that has to stay intact
            self._ring = [self.engine.host_array((n, size, size)) for _ in range(stack + max(1, pool) - (1 if obs_layout == "planes" else 0))]
            for r in self._ring:
                r[...] = 0
            self._head = 0                       # ring slot of the newest plane
            if obs_layout == "host_stack":
                self._ring = self._ring[:1]      # one landing plane is enough: the stack itself lives in the pool arrays
                self._stacked = None             # the array the last observation went out in (the roll's source)
        self._st = {"reward": self.engine.host_array((n,), np.float32), "done": self.engine.host_array((n,), np.uint8),
                    "ep_done": self.engine.host_array((n,), np.uint8), "ep_return": self.engine.host_array((n,), np.float32),
                    "ep_length": self.engine.host_array((n,), np.int32)}

    # ------------------------------------------------------------------ observation plumbing
    def _next_obs_array(self):
        if not self._pool:
            return np.empty((self.num_envs, self.size, self.size, self.stack), np.uint8)
        a = self._pool[self._turn % len(self._pool)]
        self._turn += 1
        return a

    def _landing(self):
        """(where the device's output of this step goes, keyword of the descriptor)"""
        if self.obs_layout == "device_stack":
            return self._next_obs_array(), "obs"
        if self.obs_layout == "host_stack":
            return self._ring[0], "plane"
        self._head = (self._head + 1) % len(self._ring)
        return self._ring[self._head], "plane"

    def _stacked_obs(self, landed, done, reset):
        """What VecFrameStack.step_wait / .reset (vec_frame_stack.py:17-33) -- or FrameStack inside the envs
        (atari_wrappers.py:261-275) -- hand out once the new plane has landed."""
        if self.obs_layout == "device_stack":
            return landed
        k = self.stack
        if self.obs_layout == "planes":
            R = len(self._ring)
            older = [self._ring[(self._head - j) % R] for j in range(1, k)]
            idx = slice(None) if reset else np.flatnonzero(done)
            if reset or len(idx):
                for p in older:                  # stackedobs[i] = 0  (FrameStack.reset: the observation k times)
                    p[idx] = landed[idx] if self._fill_repeat else 0
            return PlaneStack(older[::-1] + [landed])
        out = self._next_obs_array()                 # (one pool array: rolled in place)
        src = self._stacked if self._stacked is not None else out
        self.engine.host_stack_push(out, src, landed, done=None if reset else done, reset=reset, fill_repeat=self._fill_repeat)
        self._stacked = out
        return out

    def _obs(self, obs):
        # ScaledFloatFrame.observation: np.array(observation).astype(np.float32) / 255.0
        return np.asarray(obs).astype(np.float32) / 255.0 if self.scale else obs

    def reset(self):
        if self._in_flight is not None:
            self.step_wait()
        landed, key = self._landing()
        self.engine.agent_reset()
        self.engine.agent_fetch(**{key: landed})
        return self._obs(self._stacked_obs(landed, None, True))

    def step_async(self, actions):
        a = np.asarray(actions)
        if a.shape != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        if a.min() < 0 or a.max() >= len(self._action_set):
            raise AssertionError("action index out of range")
        landed, key = self._landing()
        st = self._st
        self.engine.agent_step_begin(self._lut[a.astype(np.int64)], reward=st["reward"], done=st["done"], ep_done=st["ep_done"],
                                     ep_return=st["ep_return"], ep_length=st["ep_length"], **{key: landed})
        self._in_flight = landed

    def step_wait(self):
        assert self._in_flight is not None, "step_wait without step_async"
        landed, self._in_flight = self._in_flight, None
        self.engine.agent_step_end()
        st = self._st
        reward, done = st["reward"].copy(), st["done"].astype(bool)
        ended = np.flatnonzero(st["ep_done"])
        ret, length = st["ep_return"], st["ep_length"]
        t = round(time.time() - self.tstart, 6) if len(ended) else 0.0
        extras = {int(i): {"episode": {"r": float(ret[i]), "l": int(length[i]), "t": t}} for i in ended}
        return self._obs(self._stacked_obs(landed, done, False)), reward, done, LazyInfos(self.num_envs, None, extras)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if not self.closed:
            if self._in_flight is not None:
                try:
                    self.step_wait()
                except Exception:
                    pass
            self.engine.close()
            self.closed = True
