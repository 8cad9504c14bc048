"""Agent-side preprocessing: the fused path against (a) committed fixtures recorded from the reference's
own wrapper classes (tests/golden/wrappers/, generator tests/golden/make_wrapper_golden.py) -- replayed through the CPU
restatement here and through the HIP library on the GPU box -- and (b) on the GPU box, the HIP library against the CPU
restatement on larger batches, bit for bit."""
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


# ------------------------------------------------------------------ the reference's wrapper stack, as committed fixtures
# tests/golden/wrappers/*.npz hold what the reference's OWN classes return (NoopResetEnv, MaxAndSkipEnv, bench.Monitor,
# EpisodicLifeEnv, FireResetEnv, WarpFrame, ClipRewardEnv, DummyVecEnv, VecFrameStack over ToyboxBaseEnv), recorded in the
# build container by tests/golden/make_wrapper_golden.py from /root/reference's unmodified files.  Nothing of the reference
# is restated here: the tests replay a fixture's inputs through the fused engine and compare with its outputs.
import json  # noqa: E402
import os  # noqa: E402

from conftest import GOLDEN  # noqa: E402
from support import LEGAL, amidar_edit_last_lives  # noqa: E402
from toybox_amd.toybox import codec  # noqa: E402


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, "wrappers", name + ".npz"), allow_pickle=False)
        self.name = name
        self.a = {k: z[k] for k in z.files}
        self.meta = json.loads(str(self.a["meta"]))
        self.game, self.n = self.meta["game"], self.meta["n"]
        self.legal = np.asarray(sorted(LEGAL[self.game]), np.int32)

    def __getitem__(self, k):
        return self.a[k]

    def engine(self, lib):
        """the fused engine set up from the fixture's inputs: env i seeded seed + i, the wrapper options of the case"""
        m = self.meta
        e = Engine(self.game, self.n, lib=lib)
        e.seed(m["seed"])
        e.agent_init(skip=m["skip"], out_h=m["oh"], out_w=m["ow"], stack=m["stack"], clip_reward=m["clip"],
                     episodic_life=m.get("episodic", False), fire_reset=m.get("fire", False), noop_max=m.get("noop_max", 0),
                     noop_seed=m.get("noop_seed", 0), env_offset=m.get("env_offset", 0),
                     stack_fill=1 if m.get("per_env_stack") else 0)
        return e

    def seen(self, obs):
        """what the fixture's learner saw of a uint8 observation: ScaledFloatFrame's float32 / 255 where the case has it"""
        return obs.astype(np.float32) / 255.0 if self.meta.get("scale") else obs

    def ale(self, idx):
        """the fixtures hold action INDICES (what a learner emits, envs/atari/base.py:123-126); the engine takes ALE ids"""
        return self.legal[np.asarray(idx)]

    def replay(self, e, first=0, last=None, prefix="", episodes=True):
        """steps [first, last) of the fixture through the fused engine; every output compared"""
        idx = self.a[prefix + "action_idx"]
        last = len(idx) if last is None else last
        eps = [[] for _ in range(self.n)]
        for t in range(first, last):
            obs, rew, done = e.agent_step(self.ale(idx[t]), tolerate_needs_reset=bool(prefix))
            assert np.array_equal(rew, self.a[prefix + "rew"][t]) and np.array_equal(done, self.a[prefix + "done"][t]), (self.name, t)
            assert np.array_equal(self.seen(obs), self.a[prefix + "obs"][t]), (self.name, t)
            if episodes and not prefix:
                ended, ret, length = e.agent_episodes()
                assert np.array_equal(ended, self.a["ep_flag"][t]), (self.name, t)       # info["episode"] of bench.Monitor
                assert np.array_equal(ret[ended], self.a["ep_r"][t][ended]) and np.array_equal(length[ended], self.a["ep_l"][t][ended])
                for i in np.flatnonzero(ended):
                    eps[i].append((float(ret[i]), int(length[i])))
        return eps

    def check_states(self, e, key="state_json", rng_key="sim_rng"):
        cd = codec(self.game)
        for i in range(self.n):
            assert cd.state_to_json(e.get_state(i)) == json.loads(str(self.a[key][i])), (self.name, i)
            if rng_key is not None:
                assert list(e.get_sim_rng(i)) == [int(v) for v in self.a[rng_key][i]], (self.name, i)

    def check_monitor(self, eps):
        """Monitor.episode_rewards / episode_lengths at the end of the run == the episodes the engine reported step by step"""
        pos = 0
        for i in range(self.n):
            c = int(self.a["mon_count"][i])
            want = list(zip([float(v) for v in self.a["mon_r"][pos:pos + c]], [int(v) for v in self.a["mon_l"][pos:pos + c]]))
            assert eps[i] == want, (self.name, i)
            pos += c


def test_wrapper_fixtures_are_what_the_generator_makes():
    """build container only: regenerating the fixtures from /root/reference reproduces the committed files"""
    import subprocess
    import sys
    from conftest import ROOT
    if not os.path.isdir("/root/reference/baselines/baselines/common"):
        pytest.skip("the reference tree is only present in the build container")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_wrapper_golden.py"), "--check"], cwd="/tmp",
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request, oracle_lib):
    """the library under the fused engine: the CPU restatement here, the HIP library on the GPU box"""
    if request.param == "oracle":
        return oracle_lib
    from toybox_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("game", GAMES)
def test_fused_equals_wrapper_composition(game, lib):
    """MaxAndSkip + Monitor + Warp + Clip + DummyVecEnv + VecFrameStack (no reset-time wrappers); geometries 42x42 skip 4
    stack 4, 50x40 skip 3 stack 2, 42x64 skip 1 stack 4."""
    c = Case("composition_" + game)
    e = c.engine(lib)
    assert np.array_equal(e.agent_reset(), c["reset_obs"])
    eps = c.replay(e)
    c.check_states(e)
    c.check_monitor(eps)
    if game == "breakout":
        assert c["done"].sum() > 0


@pytest.mark.parametrize("game", GAMES)
def test_fused_equals_reference_factory_functions(game, lib):
    """The reference's own make_atari + wrap_deepmind (their fixed options: TimeLimit, NoopResetEnv(30), MaxAndSkipEnv(4),
    EpisodicLifeEnv, FireResetEnv, WarpFrame 84x84, ClipRewardEnv) + Monitor + VecFrameStack(4) -- baselines' Atari path
    (atari_wrappers.py:324-360, cmd_util.py:32)."""
    c = Case("default_path_" + game)
    e = c.engine(lib)
    assert np.array_equal(e.agent_reset(), c["reset_obs"])
    eps = c.replay(e)
    c.check_states(e)
    c.check_monitor(eps)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", ["env_stack_breakout", "env_stack_space_invaders", "default_path_frame_stack_scale_breakout"])
def test_fused_equals_frame_stack_inside_every_env(name, generic, lib):
    """wrap_deepmind(frame_stack=True[, scale=True]) (atari_wrappers.py:346-360): a FrameStack(k) -- and a ScaledFloatFrame --
    inside every env and no VecFrameStack over the vector env.  A reset fills the whole stack with its observation
    (FrameStack.reset, :257-261) where VecFrameStack leaves zeros; the third case goes through the reference's own
    make_atari / wrap_deepmind functions and holds float32 observations."""
    c = Case(name)
    assert c.meta["per_env_stack"]
    e = c.engine(lib)
    if generic:                                              # two gray renders + the generic warp kernel (HIP library)
        e.set_option(_abi.OPT_AGENT_GENERIC, 1)
    first = e.agent_reset()
    assert np.array_equal(c.seen(first), c["reset_obs"])
    k = c.meta["stack"]
    assert all(np.array_equal(first[..., j], first[..., k - 1]) for j in range(k)) and first.max() > 0
    eps = c.replay(e)
    c.check_states(e)
    c.check_monitor(eps)
    assert c["done"].sum() > 0                               # resets happened inside the run as well


def test_preproc_vec_env_frame_stack_and_scale_options(oracle_lib):
    """the adapter's frame_stack="env" / scale=True against the same fixture, through ToyboxPreprocVecEnv itself"""
    from toybox_amd.envs import ToyboxPreprocVecEnv
    c = Case("default_path_frame_stack_scale_breakout")
    m = c.meta
    eng = Engine("breakout", c.n, lib=oracle_lib)
    eng.seed(m["seed"])
    env = ToyboxPreprocVecEnv("breakout", c.n, skip=m["skip"], size=m["oh"], stack=m["stack"], clip_rewards=m["clip"], engine=eng,
                              episode_life=m["episodic"], fire_reset=m["fire"], noop_max=m["noop_max"], noop_seed=m["noop_seed"],
                              env_offset=m["env_offset"], frame_stack="env", scale=True)
    assert env.observation_space.dtype == np.float32
    obs = env.reset()
    assert obs.dtype == np.float32 and np.array_equal(obs, c["reset_obs"])
    for t in range(len(c["action_idx"])):
        obs, rew, done, _ = env.step(c["action_idx"][t])
        assert np.array_equal(obs, c["obs"][t]) and np.array_equal(rew, c["rew"][t]) and np.array_equal(done, c["done"][t]), t
    with pytest.raises(ValueError):
        ToyboxPreprocVecEnv("breakout", 1, engine=eng, frame_stack="lazy")
    env.close()


ADAPTER_CASES = ["composition_breakout", "composition_space_invaders", "composition_amidar", "default_path_breakout",
                 "default_path_space_invaders", "default_path_amidar", "env_stack_breakout", "env_stack_space_invaders",
                 "default_path_frame_stack_scale_breakout", "wrappers_breakout_e1_f1_n30", "wrappers_space_invaders_e1_f1_n7",
                 "wrappers_amidar_e0_f1_n30"]


@pytest.mark.parametrize("layout", ["device_stack", "planes", "host_stack"])
@pytest.mark.parametrize("name", ADAPTER_CASES)
def test_preproc_vec_env_host_delivery_layouts(name, layout, lib):
    """What the reference's VecFrameStack / FrameStack hand a learner, through ToyboxPreprocVecEnv's three ways of getting the
    observation to the host: the whole stacks from the device; ONE new plane per env and step with the stack
    kept on the host as a ring of planes (PlaneStack) or rolled into a real array -- step_async() queues, step_wait() collects.
    Every reset() / step() output of the fixture (recorded from the reference's own classes) must come back, rotating pool,
    done-aware fill (zeros for VecFrameStack, the reset observation for FrameStack) and episode infos included."""
    from toybox_amd.envs import ToyboxPreprocVecEnv
    from toybox_amd.envs.vec_env import PlaneStack
    c = Case(name)
    m = c.meta
    if m["ow"] != m["oh"]:
        pytest.skip("the adapter takes one `size`")
    eng = Engine(c.game, c.n, lib=lib)
    eng.seed(m["seed"])
    env = ToyboxPreprocVecEnv(c.game, c.n, skip=m["skip"], size=m["oh"], stack=m["stack"], clip_rewards=m["clip"], engine=eng,
                              episode_life=m.get("episodic", False), fire_reset=m.get("fire", False), noop_max=m.get("noop_max", 0),
                              noop_seed=m.get("noop_seed", 0), env_offset=m.get("env_offset", 0),
                              frame_stack="env" if m.get("per_env_stack") else "vec", scale=bool(m.get("scale")),
                              obs_layout=layout, obs_pool=2)
    obs = env.reset()
    assert np.array_equal(np.asarray(obs), c["reset_obs"])
    kept = None
    for t in range(len(c["action_idx"])):
        env.step_async(c["action_idx"][t])
        if kept is not None and layout != "planes":          # the previous observation stays intact while the next step is in flight
            assert np.array_equal(kept[0], kept[1])
        obs, rew, done, infos = env.step_wait()
        if layout == "planes" and not m.get("scale"):
            assert isinstance(obs, PlaneStack) and obs.shape == c["obs"][t].shape
            assert np.array_equal(obs[c.n - 1], c["obs"][t][c.n - 1])
        assert np.array_equal(np.asarray(obs), c["obs"][t]), (name, layout, t)
        assert np.array_equal(rew, c["rew"][t]) and np.array_equal(done, c["done"][t]), (name, layout, t)
        ended = np.flatnonzero(c["ep_flag"][t])
        got = infos.with_key("episode")
        assert sorted(got) == [int(i) for i in ended]
        for i in ended:
            ep = got[int(i)]                                 # {"r", "l", "t"}: bench/monitor.py:64; t is wall-clock, the fixture holds r and l
            assert {k: ep[k] for k in ("r", "l")} == {"r": float(c["ep_r"][t][i]), "l": int(c["ep_l"][t][i])} and ep["t"] >= 0.0
        kept = (obs, np.asarray(obs).copy()) if not m.get("scale") else None
    c.check_states(eng)
    env.close()
    # an observation a caller kept outlives the env (page-locked memory is owned by the arrays)
    if kept is not None:
        assert np.array_equal(np.asarray(kept[0]), kept[1])


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
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("game,oh,ow,stack", [("breakout", 84, 84, 4), ("space_invaders", 84, 84, 4), ("amidar", 84, 84, 4),
                                              ("gridworld", 84, 84, 4), ("breakout", 45, 71, 3), ("amidar", 50, 41, 2)])
def test_gpu_newest_plane_and_async_host_delivery(game, oh, ow, stack, generic, hip_lib, oracle_lib):
    """tbx_agent_config_t::new_plane on the HIP library: TBX_BUF_AGENT_PLANE (what tbx_agent_step_begin delivers as `plane`) is
    the newest slot of every stack, from the fused observation kernels and from the generic warp kernel, for plane sizes that
    are and are not a multiple of four bytes; every output of the begin / end form equals the oracle's synchronous step, with
    the wrappers on, through episode ends; a second _begin without _end, an _end without _begin and a plane request without
    new_plane are TBX_E_INVALID."""
    from toybox_amd._lib import ToyboxAmdError
    # 2 048 envs and more: the fused observation kernels go out in four chunks of envs, each chunk's copy beside the next chunk's kernel
    n = 2501 if (oh == 84 and not generic and game in ("breakout", "space_invaders", "amidar")) else 300
    wrappers = game != "gridworld"
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    if generic:
        g.set_option(_abi.OPT_AGENT_GENERIC, 1)
    for e in (g, o):
        e.seed(77)
        e.agent_init(skip=3, out_h=oh, out_w=ow, stack=stack, clip_reward=True, episodic_life=wrappers, fire_reset=wrappers,
                     noop_max=9 if wrappers else 0, noop_seed=5, new_plane=True)
    og, oo = g.agent_reset(), o.agent_reset()
    assert np.array_equal(og, oo)
    plane = g.host_array((n, oh, ow))
    g.agent_fetch(plane=plane)
    assert np.array_equal(plane, oo[..., -1])
    bufs = {"reward": g.host_array((n,), np.float32), "done": g.host_array((n,), np.uint8), "obs": g.host_array((n, oh, ow, stack)),
            "ep_done": g.host_array((n,), np.uint8), "ep_return": g.host_array((n,), np.float32), "ep_length": g.host_array((n,), np.int32)}
    ends = 0
    for t in range(150 if n == 300 else 40):
        a = synthetic_actions(game, n, t, seed=3)
        g.agent_step_begin(a, plane=plane, **bufs)
        if t == 0:
            with pytest.raises(ToyboxAmdError) as ei:
                g.agent_step_begin(a, plane=plane)
            assert ei.value.code == _abi.E_INVALID
        oo, ro, do = o.agent_step(a, tolerate_needs_reset=True)   # (the CPU step runs while the device step is in flight)
        g.agent_step_end(tolerate_needs_reset=True)
        assert np.array_equal(bufs["obs"], oo) and np.array_equal(plane, oo[..., -1]), t
        assert np.array_equal(bufs["reward"], ro) and np.array_equal(bufs["done"].astype(bool), do), t
        eo = o.agent_episodes()
        assert np.array_equal(bufs["ep_done"].astype(bool), eo[0])
        assert np.array_equal(bufs["ep_return"][eo[0]], eo[1][eo[0]]) and np.array_equal(bufs["ep_length"][eo[0]], eo[2][eo[0]])
        ends += int(do.sum())
    assert ends > 0 or game == "gridworld" or n != 300        # (the big-batch cases run 60 agent steps: not every game ends an episode in them)
    with pytest.raises(ToyboxAmdError) as ei:
        g.agent_step_end()
    assert ei.value.code == _abi.E_INVALID
    g.agent_init(skip=3, out_h=oh, out_w=ow, stack=stack)
    g.agent_reset()
    with pytest.raises(ToyboxAmdError) as ei:
        g.agent_fetch(plane=plane)
    assert ei.value.code == _abi.E_INVALID
    g.close(); o.close()


RING_CASES = [("breakout", 84, 84, 4, 0), ("space_invaders", 84, 84, 4, 0), ("amidar", 84, 84, 4, 1), ("gridworld", 84, 84, 4, 0),
              ("breakout", 45, 71, 3, 1), ("amidar", 50, 41, 2, 0), ("space_invaders", 64, 64, 1, 0)]


def _ring_against_stack(ring_engine, stack_engine, game, n, oh, ow, stack, steps, wrappers):
    """every reset / agent step: the ring (TBX_BUF_AGENT_RING + tbx_agent_ring_head) read as a stack == the rolled stack of the
    other engine; TBX_BUF_AGENT_PLANE is the head slot; rewards, dones and episode records agree"""
    from support import read_buffer, stack_from_ring
    from toybox_amd._lib import ToyboxAmdError
    r, s = ring_engine, stack_engine
    assert r.agent_reset() is None
    want = s.agent_reset()
    heads = []
    ends = 0
    for t in range(-1, steps):
        if t >= 0:
            a = synthetic_actions(game, n, t, seed=9)
            o1, rew1, done1 = r.agent_step(a, tolerate_needs_reset=True)
            want, rew2, done2 = s.agent_step(a, tolerate_needs_reset=True)
            assert o1 is None and np.array_equal(rew1, rew2) and np.array_equal(done1, done2), t
            e1, e2 = r.agent_episodes(), s.agent_episodes()
            assert np.array_equal(e1[0], e2[0]) and np.array_equal(e1[1][e1[0]], e2[1][e2[0]]) and np.array_equal(e1[2][e1[0]], e2[2][e2[0]])
            ends += int(done1.sum())
        head = r.agent_ring_head()
        heads.append(head)
        ring = read_buffer(r, _abi.BUF_AGENT_RING, (stack, n, oh, ow))
        assert np.array_equal(stack_from_ring(ring, head), want), t
        assert np.array_equal(read_buffer(r, _abi.BUF_AGENT_PLANE, (n, oh, ow)), ring[head]), t
        assert r.device_buffer(_abi.BUF_AGENT_PLANE)[0] == r.device_buffer(_abi.BUF_AGENT_RING)[0] + head * n * oh * ow
    assert heads == [(heads[0] + i) % stack for i in range(len(heads))]
    assert ends > 0 or not (game == "breakout" or (game == "amidar" and steps >= 200))     # (SpaceInvaders' first life lasts longer than these rollouts)
    # there is no rolled stack in this mode: every way of asking for one is TBX_E_INVALID
    plane, obs = r.host_array((n, oh, ow)), r.host_array((n, oh, ow, stack))
    for call in (lambda: r.device_buffer(_abi.BUF_AGENT_OBS), lambda: r.agent_fetch(obs=obs),
                 lambda: r.agent_step_begin(synthetic_actions(game, n, 0), obs=obs),
                 lambda: r._check(r._lib.tbx_agent_reset(r._h, obs.ctypes.data_as(C.c_void_p))),
                 lambda: r._check(r._lib.tbx_agent_step(r._h, synthetic_actions(game, n, 0).ctypes.data_as(C.c_void_p), None, None,
                                                        obs.ctypes.data_as(C.c_void_p)))):
        with pytest.raises(ToyboxAmdError) as ei:
            call()
        assert ei.value.code == _abi.E_INVALID
    r.agent_fetch(plane=plane)
    assert np.array_equal(plane, want[..., -1])
    with pytest.raises(ToyboxAmdError) as ei:                       # ... and no ring in the other modes
        s.agent_ring_head()
    assert ei.value.code == _abi.E_INVALID
    with pytest.raises(ToyboxAmdError) as ei:
        s.device_buffer(_abi.BUF_AGENT_RING)
    assert ei.value.code == _abi.E_INVALID


@pytest.mark.parametrize("game,oh,ow,stack,fill", RING_CASES)
def test_plane_ring_is_the_rolled_stack_on_the_checker(game, oh, ow, stack, fill, oracle_lib):
    """tbx_agent_config_t::new_plane = 2 as the header states it, on the CPU restatement: its ring, read through the head index,
    holds the bytes of the rolled stack (the mode the wrapper fixtures pin: test_fixture_*), with VecFrameStack's zeroing and
    FrameStack's refill of a finished env's older frames."""
    n = 24
    wrappers = game != "gridworld"
    r, s = Engine(game, n, lib=oracle_lib), Engine(game, n, lib=oracle_lib)
    for e, mode in ((r, 2), (s, 0)):
        e.seed(31)
        e.agent_init(skip=4, out_h=oh, out_w=ow, stack=stack, clip_reward=False, episodic_life=wrappers, fire_reset=wrappers,
                     noop_max=6 if wrappers else 0, noop_seed=2, stack_fill=fill, new_plane=mode)
    _ring_against_stack(r, s, game, n, oh, ow, stack, 220, wrappers)
    r.close(); s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True], ids=["fused", "generic"])
@pytest.mark.parametrize("game,oh,ow,stack,fill", RING_CASES)
def test_gpu_plane_ring_is_the_oracles_rolled_stack(game, oh, ow, stack, fill, generic, hip_lib, oracle_lib):
    """new_plane = 2 on the HIP library (fused observation kernels and the generic warp kernel; plane sizes that are and are not
    whole 16-byte groups) against the ORACLE's rolled stack, bit for bit, through episode ends with every wrapper on."""
    if generic and (oh, ow) == (84, 84) and game != "breakout":
        pytest.skip("the generic warp kernel's ring form does not depend on the game: one 84 x 84 case and the odd sizes run it")
    n = 500
    wrappers = game != "gridworld"
    r, s = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    if generic:
        r.set_option(_abi.OPT_AGENT_GENERIC, 1)
    for e, mode in ((r, 2), (s, 0)):
        e.seed(31)
        e.agent_init(skip=4, out_h=oh, out_w=ow, stack=stack, clip_reward=False, episodic_life=wrappers, fire_reset=wrappers,
                     noop_max=6 if wrappers else 0, noop_seed=2, stack_fill=fill, new_plane=mode)
    _ring_against_stack(r, s, game, n, oh, ow, stack, 160, wrappers)
    r.close(); s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["breakout", "amidar"])
def test_gpu_plane_ring_equals_rolled_stack_at_16384_envs(game, hip_lib):
    """The two observation forms of the HIP library against each other on a batch of the size the fused kernels are tuned for
    (16 384 envs: 4 096 blocks per observation launch): after 24 agent steps with every wrapper on, the ring read through its head
    is the rolled stack, byte for byte, and the per-env outputs agree at every step."""
    from support import read_buffer, stack_from_ring
    n = 16384
    r, s = Engine(game, n, lib=hip_lib), Engine(game, n, lib=hip_lib)
    for e, mode in ((r, 2), (s, 0)):
        e.seed(2)
        e.agent_init(skip=4, clip_reward=True, episodic_life=True, fire_reset=True, noop_max=30, noop_seed=11, new_plane=mode)
    r.agent_reset()
    want = s.agent_reset()
    for t in range(24):
        a = synthetic_actions(game, n, t, seed=6)
        _, r1, d1 = r.agent_step(a, tolerate_needs_reset=True)
        want, r2, d2 = s.agent_step(a, tolerate_needs_reset=True)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        if t in (0, 11, 23):
            assert np.array_equal(stack_from_ring(read_buffer(r, _abi.BUF_AGENT_RING, (4, n, 84, 84)), r.agent_ring_head()), want), t
    r.close(); s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["breakout", "space_invaders", "amidar"])
def test_gpu_plane_ring_host_delivery_in_chunks(game, hip_lib, oracle_lib):
    """new_plane = 2 with the asynchronous host delivery of a batch big enough for the chunked form (four launches of the
    observation kernel over env ranges, each range's planes copied beside the next range's kernel): the delivered plane is the
    newest channel of the oracle's rolled stack, the device ring its whole stack, rewards / dones / episode records agree."""
    from support import read_buffer, stack_from_ring
    n = 2301
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    for e, mode in ((g, 2), (o, 0)):
        e.seed(12)
        e.agent_init(skip=4, clip_reward=True, episodic_life=True, fire_reset=True, noop_max=7, noop_seed=1, new_plane=mode)
    g.agent_reset()
    want = o.agent_reset()
    plane = [g.host_array((n, 84, 84)) for _ in range(2)]
    bufs = {"reward": g.host_array((n,), np.float32), "done": g.host_array((n,), np.uint8), "ep_done": g.host_array((n,), np.uint8)}
    for t in range(30):
        a = synthetic_actions(game, n, t, seed=8)
        g.agent_step_begin(a, plane=plane[t % 2], **bufs)
        want, ro, do = o.agent_step(a, tolerate_needs_reset=True)
        g.agent_step_end(tolerate_needs_reset=True)
        assert np.array_equal(plane[t % 2], want[..., -1]), t
        assert np.array_equal(bufs["reward"], ro) and np.array_equal(bufs["done"].astype(bool), do), t
        assert np.array_equal(bufs["ep_done"].astype(bool), o.agent_episodes()[0]), t
        if t % 7 == 0:
            assert np.array_equal(stack_from_ring(read_buffer(g, _abi.BUF_AGENT_RING, (4, n, 84, 84)), g.agent_ring_head()), want), t
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game", GAMES)
def test_gpu_step_begin_end_with_frames(game, hip_lib, oracle_lib):
    """tbx_step_begin / tbx_step_end (ToyboxVecEnv.step_async / step_wait): outputs and RGB frames of every step == the oracle's
    tbx_step + tbx_render, auto-reset on; an illegal action id is reported by the _end call like tbx_step reports it."""
    from toybox_amd._lib import ToyboxAmdError
    n = 200
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    for e in (g, o):
        e.seed(5)
        e.new_game()
    out = {"reward": g.host_array((n,), np.int32), "done": g.host_array((n,), np.uint8), "lives": g.host_array((n,), np.int32),
           "score": g.host_array((n,), np.int32)}
    frames = [g.host_array((n, g.height, g.width, 3)) for _ in range(2)]
    for t in range(120):
        a = synthetic_actions(game, n, t, seed=11)
        g.step_begin(a, auto_reset=True, frame=frames[t % 2], channels=3, **out)
        r, d, l, sc = o.step(a, auto_reset=True)
        fo = o.render(3) if t % 10 == 0 else None
        g.step_end()
        assert np.array_equal(out["reward"], r) and np.array_equal(out["done"].astype(bool), d), t
        assert np.array_equal(out["lives"], l) and np.array_equal(out["score"], sc), t
        if fo is not None:
            assert np.array_equal(frames[t % 2], fo), t
    bad = synthetic_actions(game, n, 0)
    bad[3] = 99
    g.step_begin(bad, auto_reset=True, **out)
    with pytest.raises(ToyboxAmdError) as ei:
        g.step_end()
    assert ei.value.code == _abi.E_ACTION
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game,skip,oh,ow,stack", [("breakout", 4, 84, 84, 4), ("space_invaders", 4, 84, 84, 4), ("amidar", 4, 84, 84, 4),
                                                  ("space_invaders", 3, 60, 100, 2), ("amidar", 2, 50, 40, 3), ("amidar", 1, 84, 84, 4),
                                                  ("gridworld", 4, 84, 84, 4), ("gridworld", 2, 64, 80, 1)])
def test_gpu_fused_observation_equals_generic_path(game, skip, oh, ow, stack, hip_lib):
    """The per-game fused observation kernels (Breakout: from render records; SpaceInvaders / Amidar: two painters per wave
    with class-diff scanline skipping; no full-resolution frames) == the generic render + warp path, through episode ends."""
    n = 512
    gen = Engine(game, n, lib=hip_lib)
    gen.set_option(_abi.OPT_AGENT_GENERIC, 1)
    gen.seed(77)
    gen.agent_init(skip=skip, out_h=oh, out_w=ow, stack=stack, clip_reward=False)
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
    """The whole stack: what venv.reset() / venv.step() of the reference's classes returned == the fused engine's outputs
    (observation stacks, rewards, dones, Monitor's episode records), and the games end in the same states."""
    c = Case("wrappers_%s_e%d_f%d_n%d" % (game, episodic, fire, noop_max))
    e = c.engine(lib)
    assert np.array_equal(e.agent_reset(), c["reset_obs"])
    eps = c.replay(e)
    c.check_states(e)
    c.check_monitor(eps)            # no game ended inside a reset procedure here, so info["episode"] saw them all
    n_done, n_real = int(c["done"].sum()), int(c["mon_count"].sum())
    if game == "breakout":
        assert n_done > 0 and (not episodic or n_done > n_real)


@pytest.mark.parametrize("episodic", [True, False])
def test_second_reset_in_mid_episode_is_a_noop_step_under_episodic_life(episodic, lib):
    """EpisodicLifeEnv.reset only restarts the game after a real game over (atari_wrappers.py:180-189): venv.reset() in the
    middle of an episode advances one no-op agent step.  Without the wrapper it is a real reset."""
    c = Case("second_reset_" + ("episodic" if episodic else "plain"))
    e = c.engine(lib)
    assert np.array_equal(e.agent_reset(), c["reset_obs"])
    c.replay(e, 0, 30)
    assert np.array_equal(e.agent_reset(), c["second_reset_obs"])
    c.check_states(e, "mid_state_json", "mid_sim_rng")
    c.replay(e, 30, 60)
    c.check_states(e)


def test_injected_noop_counts(lib):
    """NoopResetEnv.override_num_noops (atari_wrappers.py:115-123) per env."""
    c = Case("injected_noops")
    e = c.engine(lib)
    e.agent_set_noops(c["noop_counts"])                      # 0: keep the default rule for that env
    assert np.array_equal(e.agent_reset(), c["reset_obs"])
    c.check_states(e)
    # 3 no-op frames after the new game: SpaceInvaders' get-ready timer started at 128
    assert e.get_states_np()["life_display_timer"][0] == 128 - 3 and e.get_states_np()["life_display_timer"][3] == 127
    e.agent_set_noops(None)
    with pytest.raises(ValueError):
        e.agent_set_noops([1, 2])


def _edit_state(e, c, **kw):
    cd = codec("amidar")
    e.set_state(0, cd.state_from_json(amidar_edit_last_lives(cd.state_to_json(e.get_state(0)), **kw)))


def test_game_over_inside_the_episodic_life_noop_step(lib):
    """EpisodicLifeEnv.reset ignores the `done` of its no-op step (atari_wrappers.py:186-187).  bench.Monitor then raises on
    the next step ("Tried to step environment that needs reset", bench/monitor.py:52-53: recorded as `raised` in the
    fixture); the engine reports TBX_E_NEEDS_RESET and has carried the step out exactly as the stack does without a Monitor
    (the fixture's `cont_*` arrays, recorded from the same classes without the Monitor): the finished game reports done on
    its next frame and is reset for real."""
    from toybox_amd import ToyboxAmdError
    hits = 0
    for j in range(1, 9):
        c = Case("noop_step_game_over_j%d" % j)
        for strict in (True, False):
            e = c.engine(lib)
            assert np.array_equal(e.agent_reset(), c["reset_obs"])
            _edit_state(e, c, lives=2, jump_timer=j, perimeter_from_start=True)
            # a game that ends inside the ignored no-op step closes Monitor's episode there: info["episode"] of the step that
            # the learner sees does not carry it (ep_flag False), Monitor's own list -- and the engine's report -- do
            c.replay(e, episodes=False)
            c.check_states(e)
            ended, ret, length = e.agent_episodes()
            c.check_monitor([[(float(ret[0]), int(length[0]))] if ended[0] else []])
            assert bool(c["hit"]) == bool(c["done"][0][0] and e.get_state(0).lives == 0)
            assert bool(ended[0]) == bool(c["hit"] or c["ep_flag"][0][0])
            if not c["hit"]:
                continue
            hits += 1
            assert bool(c["raised"]) and e.agent_episodes()[0][0]
            if strict:
                with pytest.raises(ToyboxAmdError) as ei:
                    e.agent_step(c.ale([1]))
                assert ei.value.code == _abi.E_NEEDS_RESET
                continue
            for t in range(len(c["cont_action_idx"])):
                obs, rew, done = e.agent_step(c.ale(c["cont_action_idx"][t]), tolerate_needs_reset=True)
                assert np.array_equal(obs, c["cont_obs"][t]) and np.array_equal(rew, c["cont_rew"][t]), (j, t)
                assert np.array_equal(done, c["cont_done"][t]), (j, t)
                if t == 0:
                    assert done[0] and e.get_state(0).lives == 3 and not e.agent_episodes()[0][0]
            c.check_states(e, "cont_state_json", "cont_sim_rng")
    assert hits > 0


def test_cut_short_step_keeps_the_stale_frame_buffer(lib):
    """Agent steps cut short by a game over at every possible sub-frame (the jump that protects the player runs out j frames
    from now, on the last life), with FireResetEnv on: MaxAndSkipEnv stops stepping at `done` and leaves the buffer slots it
    did not reach (atari_wrappers.py:196-214); the observation is the one FireResetEnv.reset returns (:144-152)."""
    hits = 0
    cd = codec("amidar")
    for j in range(1, 14):
        c = Case("cut_short_j%d" % j)
        e = c.engine(lib)
        assert np.array_equal(e.agent_reset(), c["reset_obs"])
        _edit_state(e, c, lives=1, jump_timer=j, perimeter_from_start=False)
        for t in range(4):
            c.replay(e, t, t + 1)
            assert cd.state_to_json(e.get_state(0)) == json.loads(str(c["state_json_per_step"][t][0])), (j, t)
        hits += bool(c["done"].any())
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
    gen = Engine(game, n, lib=hip_lib, config=cfg)
    gen.set_option(_abi.OPT_AGENT_GENERIC, 1)
    gen.seed(3)
    gen.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=False, episodic_life=True, fire_reset=True, noop_max=8)
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


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_host_stack_push_is_vec_frame_stack(which, oracle_lib):
    """tbx_host_stack_push (host code of both libraries; the product's is threaded and loads without a GPU) == the lines of
    VecFrameStack.step_wait / .reset (vec_frame_stack.py:17-33) spelled with numpy, and FrameStack.reset's fill
    (atari_wrappers.py:257-261): depths 4 (dword path) and 3, in place and into a second array, one thread and several."""
    if which == "product":
        from toybox_amd import _lib
        lib = _lib.load()
    else:
        lib = oracle_lib
    rng = np.random.default_rng(5)
    for n, h, w, k, threads in ((70, 84, 84, 4, 0), (70, 84, 84, 4, 1), (9, 20, 31, 3, 3), (1, 5, 7, 4, 2), (40, 84, 84, 2, 4)):
        for fill in (0, 1):
            stacked = rng.integers(0, 256, (n, h, w, k), dtype=np.uint8)
            want = stacked.copy()
            cur = stacked.copy()
            other = np.empty_like(cur)
            for step in range(4):
                plane = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
                done = rng.random(n) < 0.3
                reset = step == 2
                # the Python: roll, clear / refill finished envs, newest frame last
                want = np.roll(want, shift=-1, axis=-1)
                fresh = np.ones(n, bool) if reset else done
                want[fresh] = plane[fresh][..., None] if fill else 0
                want[..., -1] = plane
                dst = cur if step % 2 == 0 else other           # in place, then into the second array
                rc = lib.tbx_host_stack_push(dst.ctypes.data, cur.ctypes.data, plane.ctypes.data, done.astype(np.uint8).ctypes.data,
                                             int(reset), n, h * w, k, fill, threads)
                assert rc == 0
                cur = dst
                assert np.array_equal(cur, want), (which, n, k, fill, step)
    assert lib.tbx_host_stack_push(None, None, None, None, 0, 1, 1, 1, 0, 0) == _abi.E_INVALID
