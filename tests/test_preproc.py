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


class WrapperStack:
    """numpy restatement of MaxAndSkipEnv + WarpFrame + ClipRewardEnv + VecFrameStack + VecEnv auto-reset over the
    plain per-frame engine API."""

    def __init__(self, engine, skip, oh, ow, stack, clip):
        self.e, self.skip, self.oh, self.ow, self.stack, self.clip = engine, skip, oh, ow, stack, clip
        self.obs = np.zeros((engine.n_envs, oh, ow, stack), np.uint8)

    def _warp(self, frames):
        return np.stack([area_resize_int(f[:, :, 0], self.oh, self.ow) for f in frames])

    def reset(self):
        self.e.new_game()
        self.obs[...] = 0
        self.obs[..., -1] = self._warp(self.e.render(1))
        return self.obs.copy()

    def step(self, actions):
        n = self.e.n_envs
        total = np.zeros(n, np.int64)
        fin = np.zeros(n, bool)
        buf = [None, None]
        for i in range(self.skip):
            r, d, _, _ = self.e.step(actions, auto_reset=False)
            total += np.where(fin, 0, r)
            fin |= d
            if i == self.skip - 2:
                buf[0] = self.e.render(1)
            if i == self.skip - 1:
                buf[1] = self.e.render(1)
        if fin.any():
            self.e.new_game(fin.astype(np.uint8))
            reset_frames = self.e.render(1)
        mx = buf[1] if buf[0] is None else np.maximum(buf[0], buf[1])
        if fin.any():
            mx = np.where(fin[:, None, None, None], reset_frames, mx)
        small = self._warp(mx)
        self.obs = np.roll(self.obs, -1, axis=-1)
        self.obs[fin] = 0
        self.obs[..., -1] = small
        rew = np.sign(total).astype(np.float32) if self.clip else total.astype(np.float32)
        return self.obs.copy(), rew, fin


@pytest.mark.parametrize("game,oh,ow,skip,stack,clip", [("breakout", 42, 42, 4, 4, True), ("amidar", 50, 40, 3, 2, False),
                                                       ("space_invaders", 42, 64, 1, 4, True)])
def test_fused_equals_wrapper_composition(game, oh, ow, skip, stack, clip, oracle_lib):
    n, steps = 3, 260
    fused = Engine(game, n, lib=oracle_lib)
    plain = Engine(game, n, lib=oracle_lib)
    for e in (fused, plain):
        e.seed(31)
    fused.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=clip)
    ws = WrapperStack(plain, skip, oh, ow, stack, clip)
    assert np.array_equal(fused.agent_reset(), ws.reset())
    dones = 0
    for t in range(steps):
        a = synthetic_actions(game, n, t, seed=4)
        o1, r1, d1 = fused.agent_step(a)
        o2, r2, d2 = ws.step(a)
        assert np.array_equal(d1, d2) and np.array_equal(r1, r2), t
        assert np.array_equal(o1, o2), t
        dones += int(d1.sum())
    for i in range(n):
        assert bytes(fused.get_state(i)) == bytes(plain.get_state(i))
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
# A literal, one-env-at-a-time restatement of the wrapper classes the reference's agents train under
# (baselines/baselines/common/atari_wrappers.py: NoopResetEnv :12-36, FireResetEnv :38-56, EpisodicLifeEnv :58-96,
#  MaxAndSkipEnv :99-130, ClipRewardEnv :132-139, WarpFrame :141-190; bench/monitor.py :51-76; DummyVecEnv's reset-on-done
#  vec_env/dummy_vec_env.py:45-60; VecFrameStack vec_frame_stack.py:17-30).  Two documented choices of this repo replace
# host randomness / host frames: the no-op count is a hash of (seed, global env index, episode index), and the
# observation returned by a reset is the frame after the reset procedure (not the max over its last two sub-frames).
from support import splitmix64  # noqa: E402


class _Raw:
    """ToyboxBaseEnv semantics over one env of the per-frame engine API (envs/atari/base.py:113-157)."""

    def __init__(self, eng):
        self.e = eng

    def frame(self):
        return self.e.render(1)[0, :, :, 0]

    def lives(self):
        return int(self.e.scalars()[1][0])

    def reset(self):
        self.e.new_game()
        return self.frame()

    def step(self, ale_action):
        r, d, _, _ = self.e.step(np.array([ale_action], np.int32))
        return self.frame(), int(r[0]), bool(d[0]), {}


class _NoopReset:
    def __init__(self, env, noop_max, seed, env_global):
        self.env, self.noop_max, self.seed, self.env_global, self.resets = env, noop_max, seed, env_global, 0

    def lives(self):
        return self.env.lives()

    def reset(self):
        self.resets += 1
        obs = self.env.reset()
        if self.noop_max > 0:
            k = 1 + splitmix64(self.seed ^ (self.env_global << 32) ^ self.resets) % self.noop_max
            for _ in range(k):
                obs, _, done, _ = self.env.step(0)
                if done:
                    obs = self.env.reset()
        return obs

    def step(self, a):
        return self.env.step(a)


class _MaxAndSkip:
    def __init__(self, env, skip):
        self.env, self.skip = env, skip

    def lives(self):
        return self.env.lives()

    def reset(self):
        return self.env.reset()

    def step(self, a):
        total, done, buf = 0, False, [None, None]
        for i in range(self.skip):
            obs, r, done, info = self.env.step(a)
            if i == self.skip - 2:
                buf[0] = obs
            if i == self.skip - 1:
                buf[1] = obs
            total += r
            if done:
                break
        if buf[1] is None:
            mx = obs                                # cut short by the end of the game: the caller resets anyway
        else:
            mx = buf[1] if buf[0] is None else np.maximum(buf[0], buf[1])
        return mx, total, done, info


class _Monitor:
    def __init__(self, env):
        self.env, self.rewards, self.episodes = env, [], []

    def lives(self):
        return self.env.lives()

    def reset(self):
        self.rewards = []
        return self.env.reset()

    def step(self, a):
        obs, r, done, info = self.env.step(a)
        self.rewards.append(r)
        if done:
            info = dict(info, episode={"r": float(sum(self.rewards)), "l": len(self.rewards)})
            self.episodes.append((float(sum(self.rewards)), len(self.rewards)))
        return obs, r, done, info


class _EpisodicLife:
    def __init__(self, env, on):
        self.env, self.on, self.lives_, self.was_real_done = env, on, 0, True

    def lives(self):
        return self.env.lives()

    def step(self, a):
        obs, r, done, info = self.env.step(a)
        self.was_real_done = done
        lives = self.env.lives()
        if self.on and lives < self.lives_ and lives > 0:
            done = True
        self.lives_ = lives
        return obs, r, done, info

    def reset(self):
        if self.was_real_done or not self.on:
            obs = self.env.reset()
        else:
            obs, _, d, _ = self.env.step(0)
            if d:
                obs = self.env.reset()              # this repo's rule for a game that ends inside the no-op step
        self.lives_ = self.env.lives()
        return obs


class _FireReset:
    def __init__(self, env, on, legal):
        self.env, self.on, self.legal = env, on, legal

    def reset(self):
        obs = self.env.reset()
        if self.on:
            obs, _, done, _ = self.env.step(self.legal[1])
            if done:
                self.env.reset()
            obs, _, done, _ = self.env.step(self.legal[2])
            if done:
                self.env.reset()
        return obs

    def step(self, a):
        return self.env.step(a)


def _build_stack(eng, skip, episodic, fire, noop_max, noop_seed, env_global):
    raw = _Raw(eng)
    mon = _Monitor(_MaxAndSkip(_NoopReset(raw, noop_max, noop_seed, env_global), skip))
    top = _FireReset(_EpisodicLife(mon, episodic), fire, sorted(eng.legal_actions))
    return raw, mon, top


@pytest.mark.parametrize("game,episodic,fire,noop_max", [("breakout", True, True, 30), ("breakout", True, False, 0),
                                                         ("space_invaders", True, True, 7), ("amidar", False, True, 30),
                                                         ("amidar", True, False, 5)])
def test_reset_wrappers_equal_wrapper_classes(game, episodic, fire, noop_max, oracle_lib):
    n, steps, skip, oh, ow, stack = 3, 220, 4, 42, 42, 4
    if game == "amidar":
        oh, ow = 50, 40
    if game == "space_invaders":
        oh, ow = 42, 64
    fused = Engine(game, n, lib=oracle_lib)
    fused.seed(77)
    singles = [Engine(game, 1, lib=oracle_lib) for _ in range(n)]
    for i, s in enumerate(singles):
        s.set_state(0, fused.get_state(i))
        s.set_sim_rng(fused.get_sim_rng(i), 0)
    fused.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=True, episodic_life=episodic, fire_reset=fire,
                     noop_max=noop_max, noop_seed=99, env_offset=1000)
    stacks = [_build_stack(s, skip, episodic, fire, noop_max, 99, 1000 + i) for i, s in enumerate(singles)]
    obs = np.zeros((n, oh, ow, stack), np.uint8)
    for i, (raw, mon, top) in enumerate(stacks):
        top.reset()
        obs[i, ..., -1] = area_resize_int(raw.frame(), oh, ow)
    assert np.array_equal(fused.agent_reset(), obs)
    fused_eps = [[] for _ in range(n)]
    n_done = n_real = 0
    for t in range(steps):
        a = synthetic_actions(game, n, t, seed=21)
        o1, r1, d1 = fused.agent_step(a)
        ended, ret, length = fused.agent_episodes()
        obs = np.roll(obs, -1, axis=-1)
        for i, (raw, mon, top) in enumerate(stacks):
            ob, r, done, info = top.step(int(a[i]))
            if done:
                top.reset()
                ob = raw.frame()
                obs[i] = 0
            obs[i, ..., -1] = area_resize_int(ob, oh, ow)
            assert r1[i] == float(np.sign(r)), (t, i)
            assert bool(d1[i]) == done, (t, i)
            n_done += done
            if ended[i]:
                fused_eps[i].append((float(ret[i]), int(length[i])))
        assert np.array_equal(o1, obs), t
    for i, (raw, mon, top) in enumerate(stacks):
        assert fused_eps[i] == mon.episodes, i
        assert bytes(fused.get_state(i)) == bytes(singles[i].get_state(0))
        assert fused.get_sim_rng(i) == singles[i].get_sim_rng(0)
        n_real += len(mon.episodes)
    if game == "breakout":
        assert n_done > 0 and (not episodic or n_done > n_real)


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
