"""Engine: thin Python view of one tbx_engine handle (N envs of one game on one MI355X).

All compute happens behind the C-ABI of include/toybox_amd.h; this class only marshals numpy
buffers and POD records.  The library is the in-tree HIP build (toybox_amd/_lib.py).  The `lib`
argument takes any bound library that exports the same ABI; product code never passes it (the
test-suite uses it to drive this host code over its CPU checker).
"""
import ctypes as C

import weakref

import numpy as np

from . import _abi
from ._lib import ToyboxAmdError, load


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _addr(a):
    """address of a C-contiguous numpy array for a descriptor struct (None -> 0)"""
    if a is None:
        return None
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("host output arrays must be C-contiguous")
    return a.ctypes.data


class HostArray:
    """One block of page-locked host memory from the engine library's tbx_host_alloc, handed out as a numpy array.  The block
    belongs to the ctypes array that is the `base` of every numpy view of it: a finaliser on that array calls tbx_host_free when
    the last view is gone -- by reference counting alone, no garbage-collector pass needed (the earlier owner
    object <-> buffer cycle kept blocks of several hundred MB alive until a full collection).  An observation a caller kept past
    env.close() stays valid."""

    @staticmethod
    def _release(lib, address):
        try:
            lib.tbx_host_free(C.c_void_p(address))
        except Exception:
            pass

    @staticmethod
    def make(lib, shape, dtype=np.uint8):
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = max(1, count * dt.itemsize)
        p = C.c_void_p()
        rc = lib.tbx_host_alloc(C.byref(p), nbytes)
        if rc != _abi.OK or not p.value:
            msg = lib.tbx_last_error(None)
            raise ToyboxAmdError(rc, msg.decode() if msg else "tbx_host_alloc failed")
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        weakref.finalize(buf, HostArray._release, lib, p.value)      # (holds the library and the address, not the buffer)
        return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


class Engine:
    def __init__(self, game, n_envs=1, device=0, config=None, lib=None):
        self._lib = lib if lib is not None else load()
        self.game_id = _abi.GAME_IDS[game] if isinstance(game, str) else int(game)
        self.game = _abi.GAME_NAMES[self.game_id]
        self.n_envs = int(n_envs)
        self.device = int(device)
        self._h = C.c_void_p()
        self.state_type = _abi.STATE_TYPES[self.game_id]
        self.config_type = _abi.CONFIG_TYPES[self.game_id]
        cfg_ptr, cfg_size = None, 0
        if config is not None:
            if not isinstance(config, self.config_type):
                raise TypeError("config must be a %s" % self.config_type.__name__)
            cfg_ptr, cfg_size = C.cast(C.pointer(config), C.c_void_p), C.sizeof(config)
        rc = self._lib.tbx_create(self.game_id, self.n_envs, self.device, cfg_ptr, cfg_size, C.byref(self._h))
        if rc != _abi.OK:
            msg = self._lib.tbx_last_error(None)
            self._h = C.c_void_p()
            raise ToyboxAmdError(rc, msg.decode() if msg else "tbx_create failed")
        h, w = C.c_int(), C.c_int()
        self._check(self._lib.tbx_frame_dims(self.game_id, C.byref(h), C.byref(w)))
        self.height, self.width = h.value, w.value
        buf = (C.c_int32 * 18)()
        n = self._lib.tbx_legal_actions(self.game_id, buf, 18)
        self.legal_actions = [int(buf[i]) for i in range(n)]
        self._out4 = (C.c_int32 * 4)()

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != _abi.OK:
            msg = self._lib.tbx_last_error(self._h)
            raise ToyboxAmdError(rc, msg.decode() if msg else "")

    def close(self):
        if self._h:
            self._lib.tbx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------ launch-time options (tbx_set_option)
    def set_option(self, option, value):
        self._check(self._lib.tbx_set_option(self._h, int(option), int(value)))
        return self

    def get_option(self, option):
        v = C.c_int()
        self._check(self._lib.tbx_get_option(self._h, int(option), C.byref(v)))
        return v.value

    # ------------------------------------------------------------------ seeding / RNG
    def seed(self, seed, env=-1):
        self._check(self._lib.tbx_seed(self._h, int(env), int(seed) & 0xFFFFFFFF))

    def seed_array(self, seeds):
        """env i gets seeds[i] (one upload, one launch)."""
        a = np.ascontiguousarray(seeds, dtype=np.uint32)
        if a.shape != (self.n_envs,):
            raise ValueError("seeds must have shape (%d,)" % self.n_envs)
        self._check(self._lib.tbx_seed_array(self._h, _ptr(a)))

    def get_sim_rng(self, env=0):
        out = (C.c_uint64 * 2)()
        self._check(self._lib.tbx_get_sim_rng(self._h, int(env), out))
        return int(out[0]), int(out[1])

    def set_sim_rng(self, state, env=-1):
        st = (C.c_uint64 * 2)(int(state[0]), int(state[1]))
        self._check(self._lib.tbx_set_sim_rng(self._h, int(env), st))

    # ------------------------------------------------------------------ game control
    def new_game(self, mask=None):
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert m.shape == (self.n_envs,)
        self._check(self._lib.tbx_new_game(self._h, _ptr(m)))

    def step(self, actions, auto_reset=False):
        """One frame for every env.  Returns (reward int32[N], done bool[N], lives int32[N], score int32[N])."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("actions must have shape (%d,)" % self.n_envs)
        n = self.n_envs
        reward, lives, score = (np.empty(n, np.int32) for _ in range(3))
        done = np.empty(n, np.uint8)
        flags = _abi.STEP_AUTO_RESET if auto_reset else 0
        self._check(self._lib.tbx_step(self._h, _ptr(a), flags, _ptr(reward), _ptr(done), _ptr(lives), _ptr(score)))
        return reward, done.astype(bool), lives, score

    def step1(self, env, ale_action, auto_reset=False):
        """One frame for one env: (reward, done, lives, score) as Python ints.  On a one-env engine this is the resident-kernel
        path of tbx_step1 (no launch, no copy)."""
        out = self._out4
        rc = self._lib.tbx_step1(self._h, env, ale_action, _abi.STEP_AUTO_RESET if auto_reset else 0, out)
        if rc != _abi.OK:
            self._check(rc)
        return out[0], out[1] != 0, out[2], out[3]

    def step1_frame(self, env, ale_action, channels=1, auto_reset=False):
        """tbx_step1_frame: one frame of one env and the picture of the state it leaves -- (reward, done, lives, score, frame)
        with frame a fresh uint8 (H, W, channels) array.  On a one-env engine the resident kernel paints into pinned host memory
        (no launch, no copy on the device side); the array returned here is a host copy of that buffer, because the buffer is
        reused by the next call and gym code keeps observations."""
        out = self._out4
        fp = C.c_void_p()
        rc = self._lib.tbx_step1_frame(self._h, env, ale_action, _abi.STEP_AUTO_RESET if auto_reset else 0, int(channels), out, C.byref(fp))
        if rc != _abi.OK:
            self._check(rc)
        if not fp.value:
            raise ToyboxAmdError(rc, "tbx_step1_frame returned no frame buffer")
        n = self.height * self.width * int(channels)
        frame = np.frombuffer((C.c_uint8 * n).from_address(fp.value), np.uint8).reshape(self.height, self.width, int(channels)).copy()
        return out[0], out[1] != 0, out[2], out[3], frame

    def apply_input(self, env, buttons):
        self._check(self._lib.tbx_apply_input(self._h, int(env), int(buttons)))

    def scalars(self):
        n = self.n_envs
        score, lives, level = (np.empty(n, np.int32) for _ in range(3))
        over = np.empty(n, np.uint8)
        self._check(self._lib.tbx_get_scalars(self._h, _ptr(score), _ptr(lives), _ptr(level), _ptr(over)))
        return score, lives, level, over.astype(bool)

    # ------------------------------------------------------------------ frames
    def render(self, channels=3, out=None):
        """Every env's frame as uint8 [N, H, W, channels] on the host.  `out`: write into this array instead of a fresh one
        (a reused, page-locked one -- toybox_amd.hip.PinnedArray -- receives the frames at the PCIe link's rate)."""
        shape = (self.n_envs, self.height, self.width, channels)
        if out is None:
            out = np.empty(shape, np.uint8)
        elif out.shape != shape or out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 array of shape %r" % (shape,))
        self._check(self._lib.tbx_render(self._h, _ptr(out), int(channels)))
        return out

    def render_env(self, env, channels=3):
        out = np.empty((self.height, self.width, channels), np.uint8)
        self._check(self._lib.tbx_render_env(self._h, int(env), _ptr(out), int(channels)))
        return out

    # ------------------------------------------------------------------ state / config records
    def get_state(self, env=0):
        st = self.state_type()
        self._check(self._lib.tbx_get_state(self._h, int(env), C.byref(st), C.sizeof(st)))
        return st

    def set_state(self, env, st):
        if not isinstance(st, self.state_type):
            raise TypeError("state must be a %s" % self.state_type.__name__)
        self._check(self._lib.tbx_set_state(self._h, int(env), C.byref(st), C.sizeof(st)))

    def get_states(self, first=0, count=None):
        """Records of envs [first, first+count) as a ctypes array (one pack launch, one copy)."""
        count = self.n_envs - first if count is None else int(count)
        arr = (self.state_type * count)()
        self._check(self._lib.tbx_get_states(self._h, int(first), count, C.cast(arr, C.c_void_p), C.sizeof(self.state_type)))
        return arr

    def set_states(self, first, arr):
        count = len(arr)
        if not isinstance(arr, C.Array) or arr._type_ is not self.state_type:
            raise TypeError("states must be a ctypes array of %s" % self.state_type.__name__)
        self._check(self._lib.tbx_set_states(self._h, int(first), count, C.cast(arr, C.c_void_p), C.sizeof(self.state_type)))

    def get_states_np(self, first=0, count=None):
        """The same records as a numpy structured array (fields named like the C struct): vectorised interventions,
        e.g. `st = e.get_states_np(); st['lives'] = 1; e.set_states_np(0, st)`."""
        arr = self.get_states(first, count)
        return np.frombuffer(arr, dtype=np.dtype(self.state_type)).copy()

    def set_states_np(self, first, records):
        rec = np.ascontiguousarray(records, dtype=np.dtype(self.state_type))
        arr = (self.state_type * len(rec)).from_buffer_copy(rec.tobytes())
        self.set_states(first, arr)

    def get_config(self):
        cfg = self.config_type()
        self._check(self._lib.tbx_get_config(self._h, C.byref(cfg), C.sizeof(cfg)))
        return cfg

    def set_config(self, cfg):
        if not isinstance(cfg, self.config_type):
            raise TypeError("config must be a %s" % self.config_type.__name__)
        self._check(self._lib.tbx_set_config(self._h, C.byref(cfg), C.sizeof(cfg)))

    # ------------------------------------------------------------------ batched interventions on the device
    def _edit_args(self, args):
        """args: a sequence of scalars (the same for every env) or an array [N, n_args] (one row per env)"""
        a = np.ascontiguousarray(args, dtype=np.float64)
        if a.ndim == 2:
            if a.shape[0] != self.n_envs:
                raise ValueError("per-env arguments need one row per env (%d), got %d" % (self.n_envs, a.shape[0]))
            return a, a.shape[1], 1
        a = a.reshape(-1)
        return a, a.shape[0], 0

    def edit(self, op, args=(), mask=None):
        """tbx_edit: one field write in every env whose mask entry is true (mask None: all) -- a kernel over the state in HBM,
        no state record crosses PCIe.  args as in _edit_args."""
        a, n, per_env = self._edit_args(args)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).astype(bool), dtype=np.uint8)
            if m.shape != (self.n_envs,):
                raise ValueError("mask must have one entry per env")
        self._check(self._lib.tbx_edit(self._h, int(op), _ptr(a) if n else None, n, per_env, _ptr(m) if m is not None else None))

    def reduce(self, query, args=()):
        """tbx_reduce: a per-env feature as float64 [N, width] (integers are exact; missing entries read -1)"""
        width = self._lib.tbx_reduce_width(_abi.GAME_IDS[self.game], int(query))
        if width < 0:
            raise ToyboxAmdError(width, "unknown query %d for %s" % (query, self.game))
        a, n, per_env = self._edit_args(args)
        out = np.empty((self.n_envs, width), np.float64)
        self._check(self._lib.tbx_reduce(self._h, int(query), _ptr(a) if n else None, n, per_env, _ptr(out)))
        return out

    def edit_device(self, op, args=(), mask_ptr=0, stream=0, per_env_ptr=0, n_args=0):
        """tbx_edit_device, asynchronous on `stream`: mask_ptr = device address of uint8[N] (0: every env); arguments either
        `args` (scalars, the same for every env) or per_env_ptr = device address of float64[N][n_args]"""
        if per_env_ptr:
            self._check(self._lib.tbx_edit_device(self._h, int(op), C.c_void_p(int(per_env_ptr)), int(n_args), 1,
                                                  C.c_void_p(int(mask_ptr)) if mask_ptr else None, C.c_void_p(int(stream))))
            return
        a, n, _ = self._edit_args(args)
        self._check(self._lib.tbx_edit_device(self._h, int(op), _ptr(a) if n else None, n, 0, C.c_void_p(int(mask_ptr)) if mask_ptr else None,
                                              C.c_void_p(int(stream))))

    def reduce_device(self, query, out_ptr, args=(), stream=0, per_env_ptr=0, n_args=0):
        """tbx_reduce_device: float64[N][width] into the device buffer at out_ptr, asynchronous on `stream`"""
        if per_env_ptr:
            self._check(self._lib.tbx_reduce_device(self._h, int(query), C.c_void_p(int(per_env_ptr)), int(n_args), 1, C.c_void_p(int(out_ptr)),
                                                    C.c_void_p(int(stream))))
            return
        a, n, _ = self._edit_args(args)
        self._check(self._lib.tbx_reduce_device(self._h, int(query), _ptr(a) if n else None, n, 0, C.c_void_p(int(out_ptr)), C.c_void_p(int(stream))))

    def reduce_width(self, query):
        return self._lib.tbx_reduce_width(_abi.GAME_IDS[self.game], int(query))

    def query(self, env, query_id, args, n_out=2):
        a = (C.c_int32 * len(args))(*[int(v) for v in args])
        out = (C.c_int32 * n_out)()
        self._check(self._lib.tbx_query(self._h, int(env), int(query_id), a, len(args), out, n_out))
        return [int(v) for v in out]

    # ------------------------------------------------------------------ agent-side preprocessing (fused wrapper stack)
    def agent_init(self, skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=False, fire_reset=False,
                   noop_max=0, noop_seed=0, env_offset=0, stack_fill=0, new_plane=False):
        """stack_fill: what a reset leaves in the older stack slots -- 0 zeros (VecFrameStack), 1 the reset observation (the
        per-env FrameStack of wrap_deepmind(frame_stack=True)).  new_plane = 1 / True: the device also keeps every stack's newest
        plane alone (TBX_BUF_AGENT_PLANE) -- what a host-side frame stack receives per step.  new_plane = 2: the plane INSTEAD
        of the stack -- a ring of the last `stack` planes (TBX_BUF_AGENT_RING, agent_ring_head()); agent_reset / agent_step
        return None for the observation, agent_fetch / agent_step_begin deliver plane= only."""
        cfg = _abi.AgentConfig(int(skip), int(out_h), int(out_w), int(stack), int(bool(clip_reward)), int(bool(episodic_life)),
                               int(bool(fire_reset)), int(noop_max), int(noop_seed), int(env_offset), int(stack_fill), int(new_plane))
        self._check(self._lib.tbx_agent_init(self._h, C.byref(cfg)))
        self._agent_shape = (self.n_envs, int(out_h), int(out_w), int(stack))
        self._agent_ring = int(new_plane) == 2

    def agent_ring_head(self):
        """new_plane = 2: the slot of TBX_BUF_AGENT_RING (uint8[stack][N][h][w]) that holds the newest plane; env i's stack, oldest
        first, is ring[(head + 1 + c) % stack][i], c = 0 .. stack - 1"""
        h = C.c_int32()
        self._check(self._lib.tbx_agent_ring_head(self._h, C.byref(h)))
        return int(h.value)

    def agent_set_noops(self, counts):
        """NoopResetEnv.override_num_noops per env (counts[i] > 0 overrides, 0 keeps the default rule); None removes it."""
        if counts is None:
            self._check(self._lib.tbx_agent_set_noops(self._h, None))
            return
        a = np.ascontiguousarray(counts, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("counts must have shape (%d,)" % self.n_envs)
        self._check(self._lib.tbx_agent_set_noops(self._h, _ptr(a)))

    def _agent_obs_array(self, out):
        if out is None:
            return np.empty(self._agent_shape, np.uint8)
        if out.shape != tuple(self._agent_shape) or out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 array of shape %r" % (tuple(self._agent_shape),))
        return out

    def agent_reset(self, out=None):
        obs = None if self._agent_ring else self._agent_obs_array(out)
        self._check(self._lib.tbx_agent_reset(self._h, _ptr(obs)))
        return obs

    def agent_step(self, actions, tolerate_needs_reset=False, out=None):
        """actions: ALE ids.  Returns (obs uint8[N,oh,ow,stack], reward float32[N], done bool[N]); `out`: the array the
        observations go into (a reused page-locked one takes them at the PCIe link's rate).
        TBX_E_NEEDS_RESET (where bench.Monitor raises) is raised like every other error unless tolerate_needs_reset: the step
        has been carried out either way and the outputs are valid."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("actions must have shape (%d,)" % self.n_envs)
        obs = None if self._agent_ring else self._agent_obs_array(out)
        reward = np.empty(self.n_envs, np.float32)
        done = np.empty(self.n_envs, np.uint8)
        rc = self._lib.tbx_agent_step(self._h, _ptr(a), _ptr(reward), _ptr(done), _ptr(obs))
        if not (tolerate_needs_reset and rc == _abi.E_NEEDS_RESET):
            self._check(rc)
        return obs, reward, done.astype(bool)

    def agent_episodes(self):
        """Episode monitor of the last agent step: (ended bool[N], return float32[N], length int32[N])."""
        n = self.n_envs
        ended, ret, length = np.empty(n, np.uint8), np.empty(n, np.float32), np.empty(n, np.int32)
        self._check(self._lib.tbx_agent_episodes(self._h, _ptr(ended), _ptr(ret), _ptr(length)))
        return ended.astype(bool), ret, length

    # ------------------------------------------------------------------ host delivery: step_async / step_wait
    def host_array(self, shape, dtype=np.uint8):
        """A numpy array over page-locked host memory of the engine's library (tbx_host_alloc): the destination of the
        asynchronous copies of step_begin / agent_step_begin.  The memory lives as long as the array or any view of it."""
        return HostArray.make(self._lib, shape, dtype)

    def host_stack_push(self, dst, src, plane, done=None, reset=False, fill_repeat=False, threads=0):
        """tbx_host_stack_push: dst = VecFrameStack's next stackedobs (uint8[N, h, w, stack]) from src (may be dst itself), the
        newly received plane uint8[N, h, w] and the done flags -- the roll of vec_frame_stack.py:17-27 as threaded host code"""
        n, h, w, k = dst.shape
        if src.shape != dst.shape or plane.shape != (n, h, w) or dst.dtype != np.uint8 or src.dtype != np.uint8 or plane.dtype != np.uint8:
            raise ValueError("host_stack_push: uint8 stacks [N, h, w, stack] and a uint8 plane [N, h, w] are needed")
        d = None
        if done is not None:
            d = np.ascontiguousarray(done, dtype=np.uint8)
            if d.shape != (n,):
                raise ValueError("done must have one entry per env")
        rc = self._lib.tbx_host_stack_push(_addr(dst), _addr(src), _addr(plane), _ptr(d), int(bool(reset)), n, h * w, k,
                                           int(bool(fill_repeat)), int(threads))
        if rc != _abi.OK:
            msg = self._lib.tbx_last_error(None)
            raise ToyboxAmdError(rc, msg.decode() if msg else "tbx_host_stack_push failed")
        return dst

    def agent_step_begin(self, actions, reward=None, done=None, obs=None, plane=None, ep_done=None, ep_return=None, ep_length=None):
        """tbx_agent_step_begin: queues actions -> device, the agent step and the copies of the requested outputs into the given
        host arrays (C-contiguous, of the documented dtypes; page-locked ones from host_array() keep the call asynchronous);
        returns at once.  agent_step_end() waits."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("actions must have shape (%d,)" % self.n_envs)
        out = _abi.AgentHostOut(*[_addr(x) for x in (reward, done, obs, plane, ep_done, ep_return, ep_length)])
        self._host_keep = (a, reward, done, obs, plane, ep_done, ep_return, ep_length)     # alive until the step has ended
        self._check(self._lib.tbx_agent_step_begin(self._h, _ptr(a), C.byref(out)))

    def agent_step_end(self, tolerate_needs_reset=False):
        rc = self._lib.tbx_agent_step_end(self._h)
        self._host_keep = None
        if not (tolerate_needs_reset and rc == _abi.E_NEEDS_RESET):
            self._check(rc)

    def agent_fetch(self, reward=None, done=None, obs=None, plane=None, ep_done=None, ep_return=None, ep_length=None):
        """the current contents of the agent layer's output buffers (after agent_reset or a step), synchronous"""
        out = _abi.AgentHostOut(*[_addr(x) for x in (reward, done, obs, plane, ep_done, ep_return, ep_length)])
        self._check(self._lib.tbx_agent_fetch(self._h, C.byref(out)))

    def step_begin(self, actions, auto_reset=False, reward=None, done=None, lives=None, score=None, frame=None, channels=3):
        """tbx_step_begin: one frame for every env plus (frame given) the picture of the state it leaves, queued; step_end() waits.
        reward / lives / score int32[N], done uint8[N], frame uint8[N, H, W, channels]."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("actions must have shape (%d,)" % self.n_envs)
        out = _abi.StepHostOut(_addr(reward), _addr(done), _addr(lives), _addr(score), _addr(frame), int(channels), 0)
        self._host_keep = (a, reward, done, lives, score, frame)
        self._check(self._lib.tbx_step_begin(self._h, _ptr(a), _abi.STEP_AUTO_RESET if auto_reset else 0, C.byref(out)))

    def step_end(self):
        rc = self._lib.tbx_step_end(self._h)
        self._host_keep = None
        self._check(rc)

    def agent_step_synthetic(self, action_seed, t, env_offset=0, stream=0):
        self._check(self._lib.tbx_agent_step_synthetic(self._h, int(action_seed), int(t), int(env_offset), C.c_void_p(int(stream))))

    def agent_step_device(self, actions_ptr, stream=0):
        self._check(self._lib.tbx_agent_step_device(self._h, C.c_void_p(int(actions_ptr)), C.c_void_p(int(stream))))

    # ------------------------------------------------------------------ device-resident path
    def step_device(self, actions_ptr, auto_reset=False, stream=0):
        flags = _abi.STEP_AUTO_RESET if auto_reset else 0
        self._check(self._lib.tbx_step_device(self._h, C.c_void_p(int(actions_ptr)), flags, C.c_void_p(int(stream))))

    def step_synthetic(self, action_seed, t, env_offset=0, auto_reset=True, stream=0):
        flags = _abi.STEP_AUTO_RESET if auto_reset else 0
        self._check(self._lib.tbx_step_synthetic(self._h, int(action_seed), int(t), int(env_offset), flags,
                                                 C.c_void_p(int(stream))))

    def render_device(self, out_ptr=0, channels=3, stream=0):
        self._check(self._lib.tbx_render_device(self._h, C.c_void_p(int(out_ptr)) if out_ptr else None,
                                                int(channels), C.c_void_p(int(stream))))

    def render_step_synthetic(self, action_seed, t, out_ptr=0, channels=3, env_offset=0, auto_reset=True, stream=0):
        """tbx_render_step_synthetic: the frame of the current state into out_ptr (0: TBX_BUF_FRAME) and one step with
        device-generated actions, one launch where the rasteriser reads step-written records (random-rollout loops only)."""
        flags = _abi.STEP_AUTO_RESET if auto_reset else 0
        self._check(self._lib.tbx_render_step_synthetic(self._h, C.c_void_p(int(out_ptr)) if out_ptr else None, int(channels),
                                                        int(action_seed), int(t), int(env_offset), flags, C.c_void_p(int(stream))))

    def device_identity(self):
        """tbx_device_identity: which device this engine drives -- {"ordinal", "pci" ("dddd:bb:dd"), "arch", "name", "total_memory",
        "compute_units"}; the CPU checker reports ordinal -1 and arch "cpu-oracle"."""
        d = _abi.DeviceIdentity()
        self._check(self._lib.tbx_device_identity(self._h, C.byref(d)))
        return {"ordinal": int(d.ordinal), "pci": "%04x:%02x:%02x" % (d.pci_domain & 0xFFFF, d.pci_bus & 0xFF, d.pci_device & 0xFF) if d.ordinal >= 0 else None,
                "arch": d.arch.decode(), "name": d.name.decode(), "total_memory": int(d.total_memory), "compute_units": int(d.compute_units)}

    def rollout_synthetic(self, action_seed, t0, k, channels=3, env_offset=0, auto_reset=True, stream=0):
        """tbx_rollout_synthetic: k consecutive render_step_synthetic calls (each followed by gather() under a K-step record ring, K = k)
        as one -- frames in BUF_ROLLOUT_FRAMES [k, N, H, W, C], step records in BUF_ROLLOUT_PACKED [k, stride].  Where the engine can
        (Breakout, SpaceInvaders; RGB / RGBA) the chunk's steps on an internal stream + its rasteriser launches on others (OPT_ROLLOUT_CHUNKS)."""
        flags = _abi.STEP_AUTO_RESET if auto_reset else 0
        self._check(self._lib.tbx_rollout_synthetic(self._h, int(channels), int(action_seed), int(t0), int(k), int(env_offset), flags,
                                                    C.c_void_p(int(stream))))

    def device_buffer(self, which):
        p, b = C.c_void_p(), C.c_size_t()
        self._check(self._lib.tbx_device_buffer(self._h, int(which), C.byref(p), C.byref(b)))
        return (p.value or 0), b.value

    # ------------------------------------------------------------------ multi-GPU record gather (RCCL behind the C-ABI)
    def gather_unique_id(self):
        """rank 0: bytes of a fresh communicator id, to be handed to every rank out of band."""
        buf = (C.c_uint8 * _abi.GATHER_ID_BYTES)()
        rc = self._lib.tbx_gather_unique_id(buf, _abi.GATHER_ID_BYTES)
        if rc != _abi.OK:
            msg = self._lib.tbx_last_error(None)
            raise ToyboxAmdError(rc, msg.decode() if msg else "tbx_gather_unique_id failed")
        return bytes(buf)

    def gather_init(self, nranks, rank, unique_id, records_per_rank=None):
        width = self.n_envs if records_per_rank is None else int(records_per_rank)
        buf = (C.c_uint8 * _abi.GATHER_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.tbx_gather_init(self._h, int(nranks), int(rank), width, buf, _abi.GATHER_ID_BYTES))
        k = self._lib.tbx_gather_every(self._h)
        self._gather_shape = (int(nranks), width) if k <= 1 else (int(nranks), k, width)

    def gather(self, out_ptr=0, stream=0):
        self._check(self._lib.tbx_gather(self._h, C.c_void_p(int(out_ptr)) if out_ptr else None, C.c_void_p(int(stream))))

    def gather_wait(self, stream=0):
        self._check(self._lib.tbx_gather_wait(self._h, C.c_void_p(int(stream))))

    def gather_every(self):
        """K of the record ring the communicator was initialised with (OPT_GATHER_EVERY at gather_init)"""
        return self._lib.tbx_gather_every(self._h)

    def gather_fill(self):
        """steps whose records wait in the ring for the next collective"""
        return self._lib.tbx_gather_fill(self._h)

    def gather_host(self):
        """uint64[nranks, records_per_rank] -- [nranks, K, records_per_rank] with a K-step ring -- of the last queued gather
        (blocks until it has finished)."""
        out = np.empty(self._gather_shape, np.uint64)
        self._check(self._lib.tbx_gather_host(self._h, _ptr(out)))
        return out

    def gather_nranks(self):
        """ranks the communicator spans, as the collective library itself reports it"""
        n = self._lib.tbx_gather_nranks(self._h)
        if n < 0:
            self._check(n)
        return n

    def gather_library(self):
        return (self._lib.tbx_gather_library(self._h) or b"").decode()

    def gather_reduce_max(self, value):
        v = C.c_double(float(value))
        self._check(self._lib.tbx_gather_reduce_max(self._h, C.byref(v)))
        return v.value

    def sync(self):
        self._check(self._lib.tbx_sync(self._h))
