"""The gym / VecEnv surfaces: ToyboxBaseEnv's semantics as the reference states them
(/root/reference/toybox/envs/atari/base.py:38-173) and the baselines VecEnv contract
(baselines/baselines/common/vec_env/__init__.py:26-131, dummy_vec_env.py:45-60), over the CPU restatement here and over
the HIP library on the GPU box."""
import ctypes as C

import numpy as np
import pytest

from toybox_amd import Engine, _abi
from toybox_amd import toybox as tbm
from toybox_amd.envs import (ACTION_MEANING, AmidarEnv, BreakoutEnv, ENV_IDS, SpaceInvadersEnv, ToyboxPreprocVecEnv, ToyboxVecEnv,
                             hash_seed, make)
from toybox_amd.envs.vec_env import LazyInfos


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def factory(request, oracle_lib):
    lib = oracle_lib if request.param == "oracle" else request.getfixturevalue("hip_lib")
    tbm.set_engine_factory(lambda game, n: Engine(game, n, lib=lib))
    make_engine = lambda game, n: Engine(game, n, lib=lib)
    make_engine.kind = request.param
    yield make_engine
    tbm.set_engine_factory(None)


def test_hash_seed_is_gyms():
    # gym.utils.seeding.hash_seed(seed) = int.from_bytes(sha512(str(seed))[:8], 'little'); spot values computed by hand
    import hashlib
    for s in (0, 1, 14, 2 ** 31):
        assert hash_seed(s) == int.from_bytes(hashlib.sha512(str(s).encode()).digest()[:8], "little")


def test_seed_builds_the_random_state_gym_built(factory):
    """ToyboxBaseEnv.seed (envs/atari/base.py:84-98) over gym.utils.seeding.np_random of the gym the reference targets: a
    RandomState (NoopResetEnv calls .randint on it) seeded with the 32-bit words of hash_seed(seed), zero high words left out;
    no seed = 8 bytes of entropy; the second value, < 2**31, seeds the simulator (never gym >= 0.26's Generator)."""
    from toybox_amd.envs.base import _int_list_from_bigint
    assert _int_list_from_bigint(0) == [0] and _int_list_from_bigint(5) == [5] and _int_list_from_bigint(2 ** 32) == [0, 1]
    assert _int_list_from_bigint(2 ** 64 - 1) == [2 ** 32 - 1] * 2
    env = BreakoutEnv()
    for s in (0, 5, 2 ** 40 + 3):
        first, second = env.seed(s)
        assert first == s and second == hash_seed(s + 1) % 2 ** 31
        assert isinstance(env.np_random, np.random.RandomState)
        want = np.random.RandomState()
        want.seed(_int_list_from_bigint(hash_seed(s)))
        assert [env.np_random.randint(1, 31) for _ in range(5)] == [want.randint(1, 31) for _ in range(5)]
    first, second = env.seed()
    assert 0 <= first < 2 ** 64 and 0 <= second < 2 ** 31
    with pytest.raises(ValueError):
        env.seed(-1)
    env.close()


@pytest.mark.parametrize("cls,game,dims,n_act", [(BreakoutEnv, "breakout", (160, 240), 4), (AmidarEnv, "amidar", (250, 160), 6),
                                                 (SpaceInvadersEnv, "space_invaders", (210, 320), 6)])
def test_base_env_semantics(cls, game, dims, n_act, factory):
    env = cls(grayscale=True)
    assert env.observation_space.shape == dims + (1,) and env.observation_space.dtype == np.uint8
    assert env.action_space.n == n_act and env.reward_range == (0, float("inf"))
    assert env.get_action_meanings() == list(ACTION_MEANING.values()) and len(env.get_action_meanings()) == 18
    s1, s2 = env.seed(13)
    assert s1 == 13 and s2 == hash_seed(14) % 2 ** 31                  # base.py:84-98
    obs = env.reset()
    assert obs.shape == dims + (1,) and obs.dtype == np.uint8
    assert env.cached_state is not None and env.score == 0
    total, done, steps = 0, False, 0
    rng = np.random.default_rng(0)
    while not done and steps < 6000:
        prev = env.toybox.get_score()
        obs, r, done, info = env.step(int(rng.integers(0, n_act)))
        assert r == max(env.toybox.get_score() - prev, 0)               # reward = max(score delta, 0)
        assert info["lives"] == env.toybox.get_lives()
        assert info["score"] == (0 if done else env.toybox.get_score())
        assert done == (env.toybox.get_lives() <= 0) == env.ale.game_over()
        assert ("cached_state" in info) == done                          # only on the game-over step
        total += r
        steps += 1
    with pytest.raises(AssertionError):
        env.step(n_act)                                                  # assert action_index < len(action_set)
    rgb = cls(grayscale=False)
    assert rgb.reset().shape == dims + (3,)
    assert rgb.render("rgb_array").shape == dims + (3,)
    rgba = cls(grayscale=False, alpha=True)
    o4 = rgba.reset()
    assert o4.shape == dims + (4,) and (o4[..., 3] == 255).all()
    for e in (env, rgb, rgba):
        e.close()
    assert make([k for k, v in ENV_IDS.items() if v is cls][0]).action_space.n == n_act


def test_seed_reproduces_rollout(factory):
    def rollout(seed):
        env = BreakoutEnv()
        env.seed(seed)
        env.reset()
        out = []
        for t in range(300):
            o, r, d, info = env.step(1 if t % 7 == 0 else 2 + (t % 2))
            out.append((r, d, info["lives"]))
        js = env.toybox.to_state_json()
        env.close()
        return out, js
    a, b, c = rollout(5), rollout(5), rollout(6)
    assert a == b and a[1] != c[1]


@pytest.mark.parametrize("game", ["breakout", "amidar", "space_invaders"])
def test_vec_env_contract(game, factory):
    n = 6
    # (Breakout: the default for small batches, info["cached_state"] on the game-over step through the synchronous path; the
    # other two: the asynchronous step_begin / step_end path)
    env = ToyboxVecEnv(game, n, grayscale=False, engine=factory(game, n), cache_terminal_state=None if game == "breakout" else False)
    assert env.cache_terminal_state == (game == "breakout")
    seeds = env.seed(100)
    assert [s[0] for s in seeds] == list(range(100, 100 + n)) and seeds[2][1] == hash_seed(103) % 2 ** 31   # cmd_util.py:31
    obs = env.reset()
    H, W = env.observation_space.shape[:2]
    assert obs.shape == (n, H, W, 3) and obs.dtype == np.uint8
    # the batch equals n single envs with the same seeds: obs, rewards, dones, infos, and the reset obs on done
    singles = []
    cls = {"breakout": BreakoutEnv, "amidar": AmidarEnv, "space_invaders": SpaceInvadersEnv}[game]
    for i in range(n):
        e = cls(grayscale=False)
        e.seed(100 + i)
        singles.append(e)
    ref = np.stack([e.reset() for e in singles])
    assert np.array_equal(obs, ref)
    rng = np.random.default_rng(3)
    ends = 0
    # (Breakout games end after ~600 frames of random play: the long run on the CPU checker; on the GPU box, where every single-env
    # step is a round trip, 250 frames like the others -- 70 s of the driver's 1 200 s otherwise)
    long_run = game == "breakout" and factory.kind == "oracle"
    for t in range(700 if long_run else 250):
        a = rng.integers(0, env.action_space.n, n)
        env.step_async(a)
        obs, rew, done, infos = env.step_wait()
        assert rew.dtype == np.float32 and done.dtype == bool and len(infos) == n
        for i, e in enumerate(singles):
            o, r, d, info = e.step(int(a[i]))
            if d:
                o = e.reset()                                            # DummyVecEnv: the obs of a done env is its reset obs
                ends += 1
            assert rew[i] == r and done[i] == d, (t, i)
            assert infos[i]["lives"] == info["lives"] and infos[i]["score"] == info["score"], (t, i)
            if game == "breakout":                                       # envs/atari/base.py:128-130: always on the game-over step
                assert ("cached_state" in infos[i]) == d == ("cached_state" in info), (t, i)
                if d:
                    assert infos[i]["cached_state"] == info["cached_state"], (t, i)
            assert np.array_equal(obs[i], o), (t, i)
    if long_run:
        assert ends > 0
    assert env.get_images().shape == (n, H, W, 3)
    with pytest.raises(ValueError):
        env.step(np.zeros(n + 1, np.int64))
    with pytest.raises(AssertionError):
        env.step(np.full(n, 99))
    env.close()
    for e in singles:
        e.close()


def test_render_into_a_given_array(factory):
    """Engine.render(out=...) writes every env's frame into the caller's array (shape and dtype checked)"""
    n = 5
    e = factory("amidar", n)
    e.seed(3); e.new_game()
    want = e.render(3)
    buf = np.zeros((n, e.height, e.width, 3), np.uint8)
    assert e.render(3, out=buf) is buf and np.array_equal(buf, want)
    for bad in (np.zeros((n, e.height, e.width, 4), np.uint8), np.zeros((n, e.height, e.width, 3), np.int32), buf[:, :, ::2]):
        with pytest.raises(ValueError):
            e.render(3, out=bad)
    e.close()


@pytest.mark.gpu
def test_vec_env_with_a_reused_page_locked_observation_buffer(hip_lib, oracle_lib):
    """reuse_obs_buffer=True: the same page-locked array comes back from every reset() / step(), holding what the default
    (a fresh array per call) holds -- checked against a second env over the CPU restatement"""
    n, game = 40, "space_invaders"
    a = ToyboxVecEnv(game, n, grayscale=False, engine=Engine(game, n, lib=hip_lib), seed=7, reuse_obs_buffer=True)
    b = ToyboxVecEnv(game, n, grayscale=False, engine=Engine(game, n, lib=oracle_lib), seed=7)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(1)
    for t in range(60):
        act = rng.integers(0, a.action_space.n, n)
        ra, rb = a.step(act), b.step(act)
        assert ra[0] is oa                                   # the one buffer
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]), t
    a.close(); b.close()
    # ... and the agent pipeline's observations
    a = ToyboxPreprocVecEnv(game, n, engine=Engine(game, n, lib=hip_lib), seed=7, reuse_obs_buffer=True)
    b = ToyboxPreprocVecEnv(game, n, engine=Engine(game, n, lib=oracle_lib), seed=7)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    for t in range(30):
        act = rng.integers(0, a.action_space.n, n)
        ra, rb = a.step(act), b.step(act)
        assert ra[0] is oa and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]), t
    a.close(); b.close()


def test_vec_env_cached_terminal_state(factory):
    eng = factory("breakout", 4)
    env = ToyboxVecEnv("breakout", 4, cache_terminal_state=True, engine=eng)
    env.seed(1)
    env.reset()
    for i in range(4):                                   # one life left: the next lost ball ends the game
        st = eng.get_state(i)
        st.lives = 1
        eng.set_state(i, st)
    rng = np.random.default_rng(2)
    seen = 0
    for t in range(4000):
        obs, rew, done, infos = env.step(rng.integers(0, 4, 4))
        for i in np.flatnonzero(done):
            st = infos[int(i)]["cached_state"]
            assert st["lives"] == 0 and len(st["bricks"]) == 108   # the terminal state, not the reset one
            assert infos[int(i)]["score"] == 0 and infos[int(i)]["lives"] == 0
            seen += 1
        if seen >= 2:
            break
    assert seen >= 2
    assert eng.scalars()[1].max() == 5                   # the finished envs play a fresh game
    env.close()


def test_lazy_infos_behaves_like_a_list_of_dicts():
    inf = LazyInfos(4, {"lives": np.array([3, 2, 1, 0]), "score": np.array([5, 0, 7, 0])}, {2: {"episode": {"r": 7.0, "l": 9}}})
    assert len(inf) == 4 and inf[0] == {"lives": 3, "score": 5} and inf[-1]["lives"] == 0
    assert [d.get("episode") for d in inf] == [None, None, {"r": 7.0, "l": 9}, None]
    assert inf[1:3][1]["episode"]["l"] == 9 and inf.with_key("episode") == {2: {"r": 7.0, "l": 9}}
    assert isinstance(inf[0]["lives"], int)
    with pytest.raises(IndexError):
        inf[4]


def test_preproc_vec_env_defaults(factory):
    env = ToyboxPreprocVecEnv("amidar", 3, engine=factory("amidar", 3))
    assert env.observation_space.shape == (84, 84, 4) and env.action_space.n == 6
    obs = env.reset()
    assert obs.shape == (3, 84, 84, 4) and obs.dtype == np.uint8 and obs[..., :3].max() == 0
    obs, rew, done, infos = env.step(np.array([1, 2, 3]))
    assert rew.dtype == np.float32 and set(np.unique(rew)) <= {-1.0, 0.0, 1.0} and len(infos) == 3
    assert obs[..., 2].max() > 0 and obs[..., 1].max() == 0               # the stack rolls by one frame per agent step
    env.close()


def test_gym_inheritance_and_registration_when_a_gym_is_importable(oracle_lib):
    """toybox/envs/atari/base.py:38 subclasses gym's AtariEnv and toybox/__init__.py:8-24 registers three ids on import.  gym is
    not installed here, so a child process imports the builder-authored stand-in (tests/stubs/gym): the env classes must
    derive from gym.Env, `gym.make(id)` must build them, and a gym.Wrapper stack must drive them."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import gym, numpy as np
import ctoybox                      # tests/shim: engines over the CPU restatement
import toybox_amd.envs as envs
assert sorted(envs.REGISTERED_WITH_GYM) == sorted(envs.ENV_IDS) and set(envs.ENV_IDS) <= set(gym.registry)
assert issubclass(envs.ToyboxBaseEnv, gym.Env) and type(envs.BreakoutEnv().action_space) is gym.spaces.Discrete
assert gym.registry["BreakoutToyboxNoFrameskip-v4"].nondeterministic and not gym.registry["AmidarToyboxNoFrameskip-v4"].nondeterministic

class Count(gym.Wrapper):
    def __init__(self, env):
        gym.Wrapper.__init__(self, env); self.steps = 0
    def step(self, a):
        self.steps += 1
        return self.env.step(a)

class Half(gym.ObservationWrapper):
    def observation(self, obs):
        return obs[::2, ::2]

for env_id, cls in envs.ENV_IDS.items():
    env = Half(Count(gym.wrappers.TimeLimit(gym.make(env_id))))
    assert isinstance(env.unwrapped, cls) and isinstance(env.unwrapped, gym.Env) and env.spec.id == env_id
    assert env.unwrapped.get_action_meanings()[:2] == ["NOOP", "FIRE"] and env.action_space.n == len(env.unwrapped._action_set)
    h, w, c = env.unwrapped.observation_space.shape
    assert env.reset().shape == ((h + 1) // 2, (w + 1) // 2, c)
    assert env.unwrapped.np_random.randint(1, 31) in range(1, 31)         # NoopResetEnv's draw
    for t in range(20):
        obs, r, d, info = env.step(t % env.action_space.n)
    assert env.env.steps == 20 and env.ale.lives() == info["lives"]       # attribute forwarding down to the env
    env.close()
print("ok")
'''
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "stubs"), os.path.join(ROOT, "tests", "shim"), ROOT]))
    p = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout + p.stderr)[-3000:]


@pytest.mark.parametrize("stack,size", [(1, 40), (3, 42), (4, 45)])
@pytest.mark.parametrize("frame_stack", ["vec", "env"])
def test_preproc_vec_env_layouts_agree_for_every_stack_depth(stack, size, frame_stack, factory):
    """The three ways ToyboxPreprocVecEnv gets observations to the host hold the same values for stack depths other than 4
    (byte-wise stack paths), plane sizes that are not a multiple of four bytes, one env, and every pool size -- step by step
    through episode ends, with the reset-time wrappers on."""
    from toybox_amd.envs.vec_env import PlaneStack
    game, n = "breakout", 5
    envs = {}
    for layout, pool in (("device_stack", 0), ("device_stack", 1), ("planes", 2), ("planes", 1), ("host_stack", 2), ("host_stack", 0)):
        envs[(layout, pool)] = ToyboxPreprocVecEnv(game, n, skip=3, size=size, stack=stack, seed=11, engine=factory(game, n), frame_stack=frame_stack,
                                                   episode_life=True, fire_reset=True, noop_max=5, noop_seed=2, obs_layout=layout, obs_pool=pool)
    first = {k: np.asarray(e.reset()).copy() for k, e in envs.items()}
    ref = first[("device_stack", 0)]
    assert ref.shape == (n, size, size, stack) and all(np.array_equal(v, ref) for v in first.values())
    rng = np.random.default_rng(3)
    dones = 0
    for t in range(120):
        a = rng.integers(0, 4, n)
        outs = {k: e.step(a) for k, e in envs.items()}
        o0, r0, d0, i0 = outs[("device_stack", 0)]
        for k, (o, r, d, info) in outs.items():
            assert np.array_equal(np.asarray(o), o0) and np.array_equal(r, r0) and np.array_equal(d, d0), (k, t)
            assert info.with_key("episode") == i0.with_key("episode")
            assert isinstance(o, PlaneStack) == (k[0] == "planes")
        dones += int(d0.sum())
    assert dones > 0
    one = ToyboxPreprocVecEnv(game, 1, size=size, stack=stack, seed=1, engine=factory(game, 1), obs_layout="planes")
    assert np.asarray(one.reset()).shape == (1, size, size, stack) and np.asarray(one.step([1])[0]).shape == (1, size, size, stack)
    for e in list(envs.values()) + [one]:
        e.close()
    with pytest.raises(ValueError):
        ToyboxPreprocVecEnv(game, 1, engine=factory(game, 1), obs_layout="rows")


def test_vec_env_step_async_then_wait_and_pool_rotation(factory):
    """ToyboxVecEnv: step_async() queues, step_wait() collects (vec_env/__init__.py:67-87); observations rotate through the pool, so
    the one returned by the previous step is intact after the next; rewards and dones come back as fresh arrays every step (rollout
    buffers keep references: ppo2.py:113); obs_pool = 0 hands out a new array per call."""
    game, n = "amidar", 6
    a = ToyboxVecEnv(game, n, grayscale=False, engine=factory(game, n), seed=4, obs_pool=2, cache_terminal_state=False)
    b = ToyboxVecEnv(game, n, grayscale=False, engine=factory(game, n), seed=4, obs_pool=0, cache_terminal_state=False)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(9)
    kept, rewards = None, []
    for t in range(40):
        act = rng.integers(0, a.action_space.n, n)
        a.step_async(act)
        with pytest.raises(Exception):
            a.engine.step_begin(np.zeros(n, np.int32))               # one step in flight at a time
        ra, rb = a.step_wait(), b.step(act)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
        assert ra[3][2] == rb[3][2]
        if kept is not None:
            assert np.array_equal(kept[0], kept[1]) and kept[0] is not ra[0]      # the previous observation is intact
        kept = (ra[0], ra[0].copy())
        rewards.append(ra[1])
        assert rb[0] is not ob
        ob = rb[0]
    assert len({id(r) for r in rewards}) == len(rewards)
    a.close(); b.close()
    assert np.array_equal(kept[0], kept[1])                           # ... and outlives the env


def test_cached_state_default_follows_the_batch_size(factory):
    """ToyboxBaseEnv.step attaches info["cached_state"] on every game-over step (envs/atari/base.py:128-130); the batched env does
    so by default up to ToyboxVecEnv.CACHE_TERMINAL_STATE_UP_TO envs and leaves the asynchronous path alone above."""
    small = ToyboxVecEnv("amidar", 3, engine=factory("amidar", 3))
    big = ToyboxVecEnv("amidar", ToyboxVecEnv.CACHE_TERMINAL_STATE_UP_TO + 1, engine=factory("amidar", ToyboxVecEnv.CACHE_TERMINAL_STATE_UP_TO + 1))
    forced = ToyboxVecEnv("amidar", 3, engine=factory("amidar", 3), cache_terminal_state=False)
    assert small.cache_terminal_state and not big.cache_terminal_state and not forced.cache_terminal_state
    for e in (small, big, forced):
        e.close()


def test_reset_between_step_async_and_step_wait(factory):
    """A step between step_async and step_wait is ended by whatever is called next (reset() queued more work, the next
    step_async failed with 'previous step has not been ended' and step_wait handed out the stale observation)."""
    game, n = "space_invaders", 5
    a = ToyboxVecEnv(game, n, engine=factory(game, n), seed=2, cache_terminal_state=False)
    b = ToyboxVecEnv(game, n, engine=factory(game, n), seed=2, cache_terminal_state=False)
    a.reset(); b.reset()
    act = np.array([1, 2, 3, 0, 1])
    a.step_async(act)
    oa = a.reset()                                  # ends the step in flight, then resets
    b.step(act)
    ob = b.reset()
    assert np.array_equal(oa, ob)
    for t in range(5):
        a.step_async(act)
        ra, rb = a.step_wait(), b.step(act)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
    with pytest.raises(AssertionError):
        a.step_wait()                               # nothing in flight
    # the preprocessing adapter already waited in reset(); the same sequence holds there
    pa = ToyboxPreprocVecEnv(game, n, engine=factory(game, n), seed=2)
    pb = ToyboxPreprocVecEnv(game, n, engine=factory(game, n), seed=2)
    pa.reset(); pb.reset()
    pa.step_async(act)
    pb.step(act)
    assert np.array_equal(pa.reset(), pb.reset())
    for e in (a, b, pa, pb):
        e.close()


@pytest.mark.parametrize("kind", ["step", "agent"])
def test_another_call_between_begin_and_end_ends_the_step_first(kind, factory):
    """include/toybox_amd.h, host delivery: "Any other call on the handle between _begin and _end ends the step first".  The
    outputs of the begun step arrive in the caller's buffers, the call in between acts on the state the step left, and the late
    "_end" reports the step's own result instead of delivering anything again."""
    game, n = "breakout", 7
    g, h = factory(game, n), factory(game, n)
    for e in (g, h):
        e.seed(8); e.new_game()
    acts = np.array([1, 3, 4, 0, 1, 3, 4], np.int32)
    if kind == "step":
        rew, lives, score = (g.host_array((n,), np.int32) for _ in range(3))
        done = g.host_array((n,), np.uint8)
        for t in range(3):
            g.step_begin(acts, reward=rew, done=done, lives=lives, score=score)
            if t == 1:
                g.new_game((np.arange(n) % 2).astype(np.uint8))        # another call: the step ends first, then half the envs restart
            elif t == 2:
                _ = g.scalars()
            want = h.step(acts)
            if t == 1:
                h.new_game((np.arange(n) % 2).astype(np.uint8))
            g.step_end()
            assert np.array_equal(rew, want[0]) and np.array_equal(done.astype(bool), want[1]) and np.array_equal(lives, want[2])
        with pytest.raises(Exception):
            g.step_end()                                                 # ... once
    else:
        for e in (g, h):
            e.agent_init(skip=4, out_h=84, out_w=84, stack=4)
            e.agent_reset()
        rew = g.host_array((n,), np.float32)
        done = g.host_array((n,), np.uint8)
        obs = g.host_array((n, 84, 84, 4), np.uint8)
        for t in range(3):
            g.agent_step_begin(acts, reward=rew, done=done, obs=obs)
            if t == 1:
                got = g.render(1)                                        # another call in between
            want = h.agent_step(acts)
            if t == 1:
                assert np.array_equal(got, h.render(1))
            g.agent_step_end()
            assert np.array_equal(obs, want[0]) and np.array_equal(rew, want[1]) and np.array_equal(done.astype(bool), np.asarray(want[2]).astype(bool))
        with pytest.raises(Exception):
            g.agent_step_end()
    for i in range(n):
        assert bytes(g.get_state(i)) == bytes(h.get_state(i))
    g.close(); h.close()


def test_episode_info_carries_r_l_and_t(factory):
    """bench.Monitor / VecMonitor hand out info["episode"] = {"r", "l", "t"} with t = seconds since the monitor was constructed,
    rounded to 6 places (bench/monitor.py:64, vec_monitor.py:31)."""
    import time
    game, n = "breakout", 4
    eng = factory(game, n)
    t0 = time.time()
    env = ToyboxPreprocVecEnv(game, n, engine=eng, seed=3)
    env.reset()
    for i in range(n):                                   # one life left and a ball about to leave: episodes end soon
        st = eng.get_state(i)
        st.lives = 1
        eng.set_state(i, st)
    seen = []
    for t in range(3000):
        _, _, done, infos = env.step(np.zeros(n, np.int64) + (1 if t % 7 == 0 else 0))
        seen += [infos[int(i)]["episode"] for i in np.flatnonzero(done) if "episode" in infos[int(i)]]
        if len(seen) >= 2:
            break
    assert len(seen) >= 2
    for ep in seen:
        assert set(ep) == {"r", "l", "t"} and ep["l"] >= 1
        assert 0.0 <= ep["t"] <= time.time() - t0 + 1e-3 and round(ep["t"], 6) == ep["t"]
    assert seen[-1]["t"] >= seen[0]["t"]
    env.close()
