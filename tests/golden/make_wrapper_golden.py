#!/usr/bin/env python3
"""Generator of tests/golden/wrappers/*.npz -- BUILD CONTAINER ONLY (it imports /root/reference, which does not travel).

What is recorded: what the reference's OWN agent-side wrapper classes return, imported unmodified from
    /root/reference/baselines/baselines/common/atari_wrappers.py   NoopResetEnv :108, FireResetEnv :137, EpisodicLifeEnv :157,
                                                                   MaxAndSkipEnv :193, ClipRewardEnv :221, WarpFrame :230,
                                                                   make_atari :337, wrap_deepmind :346
    /root/reference/baselines/baselines/bench/monitor.py           Monitor :12
    /root/reference/baselines/baselines/common/vec_env/dummy_vec_env.py, vec_frame_stack.py
    /root/reference/toybox/envs/atari/{base,breakout,amidar,space_invaders}.py   ToyboxBaseEnv and the three env classes
when they run over one-env engines of this repo's CPU oracle (through the `ctoybox`-named shim tests/shim/ctoybox).
Each fixture holds the inputs (action indices, seeds, wrapper options, injected no-op counts) and the reference stack's
outputs (observation stacks, rewards, dones, Monitor's episode records, the games' final state JSON and simulator RNG).
tests/test_preproc.py replays the inputs through the fused engine -- the CPU oracle in `-m "not gpu"`, the HIP library in
`-m gpu` -- and compares.

Stand-ins, all builder-authored (tests/stubs/, see its README): `gym` (Env / Wrapper attribute forwarding / spaces /
registration) and `cv2.resize(INTER_AREA)` = the area mean by definition in exact integer arithmetic (NOT OpenCV's float
code: parity with OpenCV's rounding is unpinned).  Two things are fed to the reference classes from outside:
  * `env.unwrapped._np_random`, from which NoopResetEnv draws its count (atari_wrappers.py:124), is a counter-based source
    that returns the engine's documented rule 1 + splitmix64(noop_seed ^ env << 32 ^ k) % noop_max for the k-th reset;
  * WarpFrame's `width` / `height` attributes (:233-234) are set after construction where a case wants another geometry
    than 84 x 84 (the default-path case leaves them alone).

    python tests/golden/make_wrapper_golden.py            # rewrite the fixtures
    python tests/golden/make_wrapper_golden.py --check    # regenerate in memory and compare with the committed files
"""
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TOYBOX_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "wrappers")


def _import_reference():
    """the reference's modules by their own names, without running baselines' package __init__ files (they pull in
    tensorflow / mpi4py): `baselines`, `baselines.common`, `baselines.bench` become bare namespace packages"""
    for p in (os.path.join(ROOT, "tests", "stubs"), os.path.join(ROOT, "tests", "shim"), os.path.join(ROOT, "tests"), ROOT, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True                      # /root/reference is read-only
    base = os.path.join(REF, "baselines", "baselines")
    for name, path in (("baselines", base), ("baselines.common", os.path.join(base, "common")),
                       ("baselines.bench", os.path.join(base, "bench"))):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    import baselines.common.atari_wrappers as aw
    from baselines.bench.monitor import Monitor
    from baselines.common.vec_env.dummy_vec_env import DummyVecEnv
    from baselines.common.vec_env.vec_frame_stack import VecFrameStack
    from toybox.envs.atari import AmidarEnv, BreakoutEnv, SpaceInvadersEnv
    import gym
    return types.SimpleNamespace(aw=aw, Monitor=Monitor, DummyVecEnv=DummyVecEnv, VecFrameStack=VecFrameStack, gym=gym,
                                 envs={"breakout": BreakoutEnv, "amidar": AmidarEnv, "space_invaders": SpaceInvadersEnv})


class CounterNoops:
    """stands where gym's np_random stands for NoopResetEnv: randint(1, noop_max + 1) of the k-th reset (k = 1, 2, ...)"""

    def __init__(self, noop_seed, global_env):
        self.noop_seed, self.global_env, self.k = noop_seed, global_env, 0

    def randint(self, low, high):
        from support import noop_count
        assert low == 1
        self.k += 1
        return noop_count(self.noop_seed, self.global_env, self.k, high - 1)


def build_stack(R, game, n, seed, skip=4, oh=84, ow=84, stack=4, clip=True, episodic=False, fire=False, noop_max=0, noop_seed=0,
                env_offset=0, monitor=True, factory=False, per_env_stack=False, scale=False):
    """make_atari + Monitor + wrap_deepmind + DummyVecEnv + VecFrameStack out of the reference's classes.  factory=True goes
    through the reference's own make_atari / wrap_deepmind functions (their fixed options), else the same classes in the same
    order with this case's options.  per_env_stack / scale: wrap_deepmind's frame_stack=True / scale=True -- a FrameStack(stack)
    (and a ScaledFloatFrame) inside every env, and no VecFrameStack over the vector env."""
    tops, parts = [], []
    for i in range(n):
        if factory:
            env = R.aw.make_atari({"breakout": "BreakoutToyboxNoFrameskip-v4", "amidar": "AmidarToyboxNoFrameskip-v4",
                                   "space_invaders": "SpaceInvadersToyboxNoFrameskip-v4"}[game], None)
            raw = env.unwrapped
            noop = env.env                                  # MaxAndSkipEnv(NoopResetEnv(TimeLimit(raw)))
            assert isinstance(noop, R.aw.NoopResetEnv) and noop.noop_max == noop_max and env._skip == skip
        else:
            raw = R.envs[game]()
            env, noop = raw, None
            if noop_max > 0:
                env = noop = R.aw.NoopResetEnv(env, noop_max=noop_max)
            env = R.aw.MaxAndSkipEnv(env, skip=skip)
        raw.toybox.set_seed(seed + i)
        raw._np_random = CounterNoops(noop_seed, env_offset + i)
        mon = None
        if monitor:
            env = mon = R.Monitor(env, None, allow_early_resets=True)
        if factory:
            env = R.aw.wrap_deepmind(env, episode_life=episodic, clip_rewards=clip, frame_stack=per_env_stack, scale=scale)
            assert fire and (stack == 4 or not per_env_stack)
        else:
            if episodic:
                env = R.aw.EpisodicLifeEnv(env)
            if fire:
                env = R.aw.FireResetEnv(env)
            env = warp = R.aw.WarpFrame(env)
            if (oh, ow) != (84, 84):
                warp.height, warp.width = oh, ow
                warp.observation_space = R.gym.spaces.Box(low=0, high=255, shape=(oh, ow, 1), dtype=np.uint8)
            if scale:
                env = R.aw.ScaledFloatFrame(env)
            if clip:
                env = R.aw.ClipRewardEnv(env)
            if per_env_stack:
                env = R.aw.FrameStack(env, stack)
        tops.append(env)
        parts.append(types.SimpleNamespace(raw=raw, noop=noop, monitor=mon))
    venv = R.DummyVecEnv([(lambda e=e: e) for e in tops])
    if not per_env_stack:
        venv = R.VecFrameStack(venv, stack)
    return venv, parts


def action_indices(game, n, t, seed):
    from support import LEGAL, synthetic_actions
    legal = np.asarray(sorted(LEGAL[game]), np.int32)
    return np.searchsorted(legal, synthetic_actions(game, n, t, seed=seed)).astype(np.int32)


class Recorder:
    def __init__(self, venv, parts):
        self.venv, self.parts, self.n = venv, parts, len(parts)
        self.idx, self.obs, self.rew, self.done, self.ep_flag, self.ep_r, self.ep_l = [], [], [], [], [], [], []

    def reset(self):
        return self.venv.reset().copy()

    def step(self, idx):
        obs, rew, done, infos = self.venv.step(np.asarray(idx))
        flag = np.array(["episode" in info for info in infos])
        self.idx.append(np.asarray(idx, np.int32))
        self.obs.append(obs.copy()); self.rew.append(rew.astype(np.float32)); self.done.append(done.astype(bool))
        self.ep_flag.append(flag)
        self.ep_r.append(np.array([info["episode"]["r"] if f else 0.0 for info, f in zip(infos, flag)], np.float32))
        self.ep_l.append(np.array([info["episode"]["l"] if f else 0 for info, f in zip(infos, flag)], np.int32))
        return obs, rew, done, infos

    def finals(self):
        """the games' state JSON and simulator RNG words, the Monitors' episode lists"""
        out = {"state_json": np.array([json.dumps(p.raw.toybox.to_state_json(), sort_keys=True) for p in self.parts]),
               "sim_rng": np.array([p.raw.toybox.config_to_json()["rand"]["state"] for p in self.parts], np.uint64),
               "lives": np.array([p.raw.toybox.get_lives() for p in self.parts], np.int32)}
        if self.parts[0].monitor is not None:
            out["mon_count"] = np.array([len(p.monitor.episode_rewards) for p in self.parts], np.int32)
            out["mon_r"] = np.array([r for p in self.parts for r in p.monitor.episode_rewards], np.float32)
            out["mon_l"] = np.array([l for p in self.parts for l in p.monitor.episode_lengths], np.int32)
        return out

    def arrays(self):
        def st(x, dt, shape):
            return np.stack(x) if x else np.zeros((0,) + shape, dt)
        n = self.n
        shape = self.venv.stackedobs.shape if hasattr(self.venv, "stackedobs") else (n,) + tuple(self.venv.observation_space.shape)
        d = {"action_idx": st(self.idx, np.int32, (n,)), "obs": st(self.obs, self.venv.observation_space.dtype, shape),
             "rew": st(self.rew, np.float32, (n,)), "done": st(self.done, bool, (n,)), "ep_flag": st(self.ep_flag, bool, (n,)),
             "ep_r": st(self.ep_r, np.float32, (n,)), "ep_l": st(self.ep_l, np.int32, (n,))}
        d.update(self.finals())
        return d


HEADER = ("outputs of the reference's own wrapper classes (atari_wrappers.py, bench/monitor.py, vec_env/*.py, "
          "toybox/envs/atari/*.py) over one-env CPU-oracle engines; cv2.resize(INTER_AREA) is a by-definition stand-in, "
          "gym is a builder-authored stub; generator: tests/golden/make_wrapper_golden.py")


def rollout_case(R, name, game, n, steps, seed, act_seed, **opt):
    venv, parts = build_stack(R, game, n, seed, **opt)
    rec = Recorder(venv, parts)
    d = {"reset_obs": rec.reset()}
    for t in range(steps):
        rec.step(action_indices(game, n, t, act_seed))
    d.update(rec.arrays())
    meta = dict(opt, game=game, n=n, steps=steps, seed=seed, act_seed=act_seed, header=HEADER)
    return name, d, meta


def second_reset_case(R, episodic):
    """venv.reset() in the middle of an episode: under EpisodicLifeEnv a no-op agent step (atari_wrappers.py:180-189)"""
    game, n = "breakout", 2
    opt = dict(skip=4, oh=42, ow=42, stack=2, clip=True, episodic=episodic, fire=True)
    venv, parts = build_stack(R, game, n, 5, **opt)
    rec = Recorder(venv, parts)
    d = {"reset_obs": rec.reset()}
    for t in range(30):
        rec.step(action_indices(game, n, t, 2))
    d["second_reset_obs"] = rec.reset()
    mid = rec.finals()
    d["mid_state_json"], d["mid_sim_rng"] = mid["state_json"], mid["sim_rng"]
    for t in range(30, 60):
        rec.step(action_indices(game, n, t, 2))
    d.update(rec.arrays())
    return "second_reset_%s" % ("episodic" if episodic else "plain"), d, dict(opt, game=game, n=n, seed=5, act_seed=2, header=HEADER)


def injected_noops_case(R):
    """NoopResetEnv.override_num_noops per env (atari_wrappers.py:115-123); 0 = leave that env on the default rule"""
    game, n, counts = "space_invaders", 4, [3, 0, 17, 1]
    opt = dict(skip=2, oh=42, ow=64, stack=1, clip=False, noop_max=30, noop_seed=1)
    venv, parts = build_stack(R, game, n, 8, **opt)
    for p, c in zip(parts, counts):
        p.noop.override_num_noops = c if c > 0 else None
    rec = Recorder(venv, parts)
    d = {"reset_obs": rec.reset(), "noop_counts": np.array(counts, np.int32)}
    d.update(rec.arrays())
    return "injected_noops", d, dict(opt, game=game, n=n, seed=8, header=HEADER)


def _edit(parts, **kw):
    from support import amidar_edit_last_lives
    tb = parts[0].raw.toybox
    tb.write_state_json(amidar_edit_last_lives(tb.to_state_json(), **kw))


def noop_step_game_over_cases(R):
    """Amidar on two lives: the first goes when the jump runs out (frame j of the agent step), everyone respawns on the
    player's tile and the next frame costs the last.  When the first life goes in the LAST frame of the agent step, the game
    ends inside EpisodicLifeEnv.reset's no-op step, whose `done` the class ignores (atari_wrappers.py:186-187): Monitor then
    raises on the next step (bench/monitor.py:52-53).  `cont_*`: the same stack WITHOUT a Monitor carries on -- done at
    once, real reset, play on."""
    out = {}
    opt = dict(skip=4, oh=50, ow=40, stack=2, clip=True, episodic=True)
    hits = 0
    for j in range(1, 9):
        runs = {}
        for monitor in (True, False):
            venv, parts = build_stack(R, "amidar", 1, 3, monitor=monitor, **opt)
            rec = Recorder(venv, parts)
            reset_obs = rec.reset()
            _edit(parts, lives=2, jump_timer=j, perimeter_from_start=True)
            rec.step([0])
            runs[monitor] = (rec, parts, reset_obs)
        rec, parts, reset_obs = runs[True]
        d = {"reset_obs": reset_obs}
        hit = bool(rec.done[0][0] and parts[0].raw.toybox.get_lives() == 0 and parts[0].monitor.needs_reset)
        first = rec.arrays()
        raised = False
        if hit:
            hits += 1
            try:
                rec.venv.step(np.array([1]))
            except RuntimeError as e:
                raised = "needs reset" in str(e)
            assert raised
            rec2, parts2, _ = runs[False]
            for t in range(12):
                rec2.step([1 + t % 3])
            cont = rec2.arrays()
            for k in ("action_idx", "obs", "rew", "done"):
                d["cont_" + k] = cont[k][1:]            # [0] is the step both stacks made before the game ended
            for k in ("state_json", "sim_rng", "lives"):
                d["cont_" + k] = cont[k]
        d.update(first)
        d["hit"], d["raised"] = np.array(hit), np.array(raised)
        out["noop_step_game_over_j%d" % j] = (d, dict(opt, game="amidar", n=1, seed=3, jump_timer=j, header=HEADER))
    assert 0 < hits < 8
    return [(k, v[0], v[1]) for k, v in out.items()]


def cut_short_cases(R):
    """agent steps cut short by the game over at every sub-frame, FireResetEnv on: MaxAndSkipEnv stops at `done` and leaves
    the buffer slots it did not reach (atari_wrappers.py:196-214); the observation is FireResetEnv.reset's (:144-152)"""
    out = []
    opt = dict(skip=4, oh=50, ow=40, stack=2, clip=False, episodic=False, fire=True)
    hits = 0
    for j in range(1, 14):
        venv, parts = build_stack(R, "amidar", 1, 4, **opt)
        rec = Recorder(venv, parts)
        d = {"reset_obs": rec.reset()}
        _edit(parts, lives=1, jump_timer=j, perimeter_from_start=False)
        states = []
        for t in range(4):
            rec.step([0])
            states.append(rec.finals()["state_json"])
        d.update(rec.arrays())
        d["state_json_per_step"] = np.stack(states)
        hits += bool(np.stack(rec.done).any())
        out.append(("cut_short_j%d" % j, d, dict(opt, game="amidar", n=1, seed=4, jump_timer=j, header=HEADER)))
    assert hits > 0
    return out


COMPOSITION = [("breakout", dict(oh=42, ow=42, skip=4, stack=4, clip=True)), ("amidar", dict(oh=50, ow=40, skip=3, stack=2, clip=False)),
               ("space_invaders", dict(oh=42, ow=64, skip=1, stack=4, clip=True))]
# (game, episodic, fire, noop_max)
WRAPPER_CASES = [("breakout", True, True, 30), ("breakout", True, False, 0), ("breakout", False, True, 4),
                 ("space_invaders", True, True, 7), ("amidar", False, True, 30), ("amidar", True, False, 5)]
GEOMETRY = {"breakout": (42, 42), "amidar": (50, 40), "space_invaders": (42, 64)}


def all_cases(R):
    cases = []
    for game, opt in COMPOSITION:
        cases.append(rollout_case(R, "composition_%s" % game, game, 3, 260, 31, 4, **opt))
    for game, episodic, fire, noop_max in WRAPPER_CASES:
        oh, ow = GEOMETRY[game]
        cases.append(rollout_case(R, "wrappers_%s_e%d_f%d_n%d" % (game, episodic, fire, noop_max), game, 3, 220, 77, 21, skip=4,
                                  oh=oh, ow=ow, stack=4, clip=True, episodic=episodic, fire=fire, noop_max=noop_max, noop_seed=99,
                                  env_offset=1000))
    # the reference's own factory functions with their fixed options: make_atari (TimeLimit, NoopResetEnv(30), MaxAndSkipEnv(4)),
    # wrap_deepmind (EpisodicLife, FireReset, WarpFrame 84x84, ClipReward), VecFrameStack(4)
    for game in ("breakout", "space_invaders", "amidar"):
        cases.append(rollout_case(R, "default_path_%s" % game, game, 2, 120, 1234, 1337, skip=4, oh=84, ow=84, stack=4, clip=True,
                                  episodic=True, fire=True, noop_max=30, noop_seed=7, env_offset=64, factory=True))
    # wrap_deepmind(frame_stack=True[, scale=True]): FrameStack (and ScaledFloatFrame) inside every env, no VecFrameStack
    cases.append(rollout_case(R, "env_stack_breakout", "breakout", 3, 200, 77, 21, skip=4, oh=42, ow=42, stack=4, clip=True, episodic=True,
                              fire=True, noop_max=5, noop_seed=99, env_offset=10, per_env_stack=True))
    cases.append(rollout_case(R, "env_stack_space_invaders", "space_invaders", 2, 160, 78, 22, skip=4, oh=42, ow=64, stack=3, clip=True,
                              episodic=True, fire=True, per_env_stack=True))
    cases.append(rollout_case(R, "default_path_frame_stack_scale_breakout", "breakout", 2, 60, 1234, 1337, skip=4, oh=84, ow=84, stack=4,
                              clip=True, episodic=True, fire=True, noop_max=30, noop_seed=7, env_offset=64, factory=True,
                              per_env_stack=True, scale=True))
    cases.append(second_reset_case(R, True))
    cases.append(second_reset_case(R, False))
    cases.append(injected_noops_case(R))
    cases += noop_step_game_over_cases(R)
    cases += cut_short_cases(R)
    return cases


def pack(d, meta):
    buf = io.BytesIO()
    np.savez_compressed(buf, meta=np.array(json.dumps(meta, sort_keys=True)), **d)
    return buf.getvalue()


def load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def main():
    check = "--check" in sys.argv
    R = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    bad = 0
    total = 0
    for name, d, meta in all_cases(R):
        path = os.path.join(OUT, name + ".npz")
        if check:
            old = load(path)
            new = dict(d, meta=np.array(json.dumps(meta, sort_keys=True)))
            same = set(old) == set(new) and all(np.array_equal(old[k], np.asarray(new[k])) for k in old)
            print("%-44s %s" % (name, "same" if same else "DIFFERENT"))
            bad += not same
        else:
            blob = pack(d, meta)
            with open(path, "wb") as f:
                f.write(blob)
            total += len(blob)
            print("%-44s %8d bytes" % (name, len(blob)))
    if not check:
        print("total %d bytes" % total)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
