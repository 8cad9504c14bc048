"""GridWorld: the reference's two golden dumps, this repo's rules through every surface, and on the GPU
box the HIP engine against the CPU restatement bit for bit."""
import json
import os

import numpy as np
import pytest

from support import synthetic_actions
from toybox_amd import Engine, _abi
from toybox_amd.games import gridworld as gw

GOLD = os.path.join(os.path.dirname(__file__), "golden")
UP, RIGHT, LEFT, DOWN = 2, 3, 4, 5


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _resolved(js):
    """grid of tile records instead of tile indices (the dumps' index order is a hash-map order)."""
    return [[json.dumps(js["tiles"][v], sort_keys=True) for v in row] for row in js["grid"]]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request, oracle_lib):
    if request.param == "oracle":
        return oracle_lib
    from toybox_amd import _lib
    return _lib.load()


def test_golden_dumps_round_trip_and_defaults(lib):
    gc, gs = _load("gridworld_config.json"), _load("gridworld_state.json")
    assert gw.config_to_json(gw.config_from_json(gc)) == gc
    assert gw.state_to_json(gw.state_from_json(gs)) == gs
    assert gw.default_config() == gc
    with Engine("gridworld", 2, lib=lib) as e:
        assert gw.config_to_json(e.get_config()) == gc           # library-side defaults
        js = gw.state_to_json(e.get_state(1))
        # a new game of the default config is the golden state up to the order of the tile table
        assert _resolved(js) == _resolved(gs)
        for k in ("score", "player", "player_color", "game_over"):
            assert js[k] == gs[k]
        assert json.dumps(js["tiles"][js["reward_becomes"]], sort_keys=True) == \
            json.dumps(gs["tiles"][gs["reward_becomes"]], sort_keys=True)
        assert (e.height, e.width) == (128, 160) and e.legal_actions == [0, 2, 3, 4, 5]
        e.set_state(0, gw.state_from_json(gs))                    # the golden state itself loads and plays
        e.step(np.array([RIGHT, RIGHT], np.int32))
        assert gw.state_to_json(e.get_state(0))["player"] == [3, 4]


def test_rules(oracle_lib):
    with Engine("gridworld", 1, lib=oracle_lib) as e:
        def go(a):
            r, d, lives, score = e.step(np.array([a], np.int32))
            return int(r[0]), bool(d[0]), gw.state_to_json(e.get_state(0))
        r, d, js = go(LEFT)
        assert js["player"] == [1, 4]
        r, d, js = go(LEFT)
        assert js["player"] == [1, 4] and r == 0                 # wall
        for a in (UP, UP, UP):
            r, d, js = go(a)
        assert js["player"] == [1, 1]
        r, d, js = go(UP)
        assert js["player"] == [1, 1]                             # border wall
        for a in (RIGHT, RIGHT):
            go(a)
        r, d, js = go(RIGHT)                                       # the reward cell at (4, 1)
        assert js["player"] == [4, 1] and r == 1 and js["score"] == 1 and not d
        assert json.dumps(js["tiles"][js["grid"][1][4]], sort_keys=True) == \
            json.dumps(js["tiles"][js["reward_becomes"]], sort_keys=True)      # collected
        r, d, js = go(LEFT)
        r, d, js = go(RIGHT)
        assert r == 0 and js["score"] == 1                         # nothing left to collect
        r, d, js = go(0)
        assert js["player"] == [4, 1]
        # walk to the goal: right to (7,1), down to (7,3), left to (5,3), down (5,4) [reward], down (5,5), right (6,5), (7,5)
        total = 0
        for a in (RIGHT, RIGHT, RIGHT, DOWN, DOWN, LEFT, LEFT, DOWN, DOWN, RIGHT, RIGHT):
            r, d, js = go(a)
            total += r
        assert js["player"] == [7, 5] and d and js["game_over"] and total == 11 and js["score"] == 12
        score, lives, level, over = e.scalars()
        assert (int(score[0]), int(lives[0]), int(level[0]), bool(over[0])) == (12, 0, 1, True)
        r, d, js2 = go(LEFT)
        assert js2["player"] == [7, 5] and d                       # frozen after the goal
        r, d, _, _ = e.step(np.array([0], np.int32), auto_reset=True)
        js3 = gw.state_to_json(e.get_state(0))
        assert bool(d[0]) and js3["player"] == [2, 4] and js3["score"] == 0 and not js3["game_over"]


def test_picture(oracle_lib):
    with Engine("gridworld", 1, lib=oracle_lib) as e:
        f = e.render(3)[0]
        tw, th = 160 // 9, 128 // 7
        assert tuple(f[0, 0]) == (0, 0, 0)                         # wall
        assert tuple(f[th + 1, tw + 1]) == (255, 255, 255)         # floor
        assert tuple(f[th + 1, 4 * tw + 1]) == (255, 255, 0)       # reward
        assert tuple(f[5 * th + 1, 7 * tw + 1]) == (0, 255, 0)     # goal
        assert tuple(f[4 * th + 1, 2 * tw + 1]) == (255, 0, 0)     # player
        assert tuple(f[127, 159]) == (0, 0, 0) and tuple(f[10, 9 * tw]) == (0, 0, 0)   # outside the board
        g = e.render(1)[0, :, :, 0]
        assert g[4 * th + 1, 2 * tw + 1] == (77 * 255 + 128) >> 8
        a = e.render(4)[0]
        assert (a[..., 3] == 255).all() and np.array_equal(a[..., :3], f)


def _maze_config(w, h, seed):
    rng = np.random.default_rng(seed)
    names = ["0", "1", "G", "R", "P"]
    rows = []
    for y in range(h):
        rows.append("".join(names[int(rng.choice(5, p=[0.55, 0.2, 0.03, 0.17, 0.05]))] for _ in range(w)))
    rows[0] = "0" + rows[0][1:]
    cfg = gw.default_config()
    cfg["tiles"]["P"] = {"color": {"r": 90, "g": 20, "b": 200, "a": 255}, "goal": False, "reward": -3, "walkable": True}
    cfg.update(game_size=[w, h], grid=rows, player_start=[0, 0], reward_becomes="0")
    return cfg


def test_toybox_and_env_surface(oracle_lib):
    from toybox_amd import toybox as tbm
    from toybox_amd.envs import GridWorldEnv
    tbm.set_engine_factory(lambda game, n: Engine(game, n, lib=oracle_lib))
    try:
        with tbm.Toybox("gridworld") as tb:
            assert tb.get_legal_action_set() == [0, 2, 3, 4, 5]
            assert tb.config_to_json() == _load("gridworld_config.json")
            inp = tbm.Input()
            inp.set_input("right")
            tb.apply_action(inp)
            assert tb.state_to_json()["player"] == [3, 4]
            assert tb.query_state_json("xy") == [3, 4]
            tb.write_config_json(_maze_config(20, 11, 3))
            js = tb.state_to_json()
            assert len(js["grid"]) == 11 and len(js["grid"][0]) == 20 and js["player"] == [0, 0] and len(js["tiles"]) == 5
            js["player"] = [5, 5]
            tb.write_state_json(js)
            assert tb.state_to_json() == js
            assert tb.get_rgb_frame().shape == (128, 160, 3)
        env = GridWorldEnv()
        obs = env.reset()
        assert obs.shape == (128, 160, 1) and env.action_space.n == 5
        obs, r, done, info = env.step(2)                           # index 2 of [0,2,3,4,5] = RIGHT
        assert info["lives"] == 1 and not done
        env.close()
    finally:
        tbm.set_engine_factory(None)


def test_limits_are_reported(oracle_lib):
    with pytest.raises(ValueError):
        gw.config_from_json(dict(gw.default_config(), game_size=[40, 7]))
    with Engine("gridworld", 1, lib=oracle_lib) as e:
        st = e.get_state(0)
        st.width = 33
        with pytest.raises(Exception) as ei:
            e.set_state(0, st)
        assert ei.value.code == _abi.E_UNSUPPORTED


# ------------------------------------------------------------------ GPU parity

def _pair(n, hip_lib, oracle_lib, config=None):
    cfg = gw.config_from_json(config) if config is not None else None
    g, o = Engine("gridworld", n, lib=hip_lib, config=cfg), Engine("gridworld", n, lib=oracle_lib, config=cfg)
    return g, o


def _same_states(g, o, envs):
    for i in envs:
        assert bytes(g.get_state(int(i))) == bytes(o.get_state(int(i))), i


@pytest.mark.gpu
@pytest.mark.parametrize("config", [None, (20, 11, 3), (32, 32, 9), (1, 1, 0), (5, 32, 2)])
def test_gpu_rollout_and_frames(config, hip_lib, oracle_lib):
    n, steps = 2048, 1200
    cfg = None if config is None else _maze_config(*config)
    g, o = _pair(n, hip_lib, oracle_lib, cfg)
    _same_states(g, o, range(0, n, 97))
    dones = 0
    for t in range(steps):
        a = synthetic_actions("gridworld", n, t)
        rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)
        for x, y, name in zip(rg, ro, ("reward", "done", "lives", "score")):
            assert np.array_equal(x, y), (name, t)
        dones += int(rg[1].sum())
        if t in (0, 5, 300, steps - 1):
            for ch in (1, 3, 4):
                fg, fo = g.render(ch), o.render(ch)
                assert np.array_equal(fg, fo), (t, ch)
    _same_states(g, o, range(n))
    for x, y in zip(g.scalars(), o.scalars()):
        assert np.array_equal(x, y)
    if config in (None, (20, 11, 3)):
        assert dones > 0
    # in-kernel actions
    for t in range(100):
        g.step_synthetic(1337, t, env_offset=11, auto_reset=True)
        o.step(synthetic_actions("gridworld", n, t, seed=1337, env_offset=11), auto_reset=True)
    g.sync()
    _same_states(g, o, range(0, n, 5))


@pytest.mark.gpu
def test_gpu_state_writes_and_config_swap(hip_lib, oracle_lib):
    n = 16
    g, o = _pair(n, hip_lib, oracle_lib)
    js = gw.state_to_json(o.get_state(3))
    js["player"] = [7, 3]
    js["tiles"].append({"color": {"r": 1, "g": 2, "b": 3, "a": 255}, "goal": False, "reward": 5, "walkable": True})
    js["grid"][3][6] = len(js["tiles"]) - 1
    js["player_color"] = {"r": 9, "g": 200, "b": 40, "a": 255}
    for e in (g, o):
        e.set_state(3, gw.state_from_json(js))
    assert gw.state_to_json(g.get_state(3)) == js
    a = np.full(n, LEFT, np.int32)
    rg, ro = g.step(a), o.step(a)
    assert rg[0][3] == 5 and np.array_equal(rg[0], ro[0])
    assert np.array_equal(g.render(3), o.render(3))
    recs = o.get_states()
    g.set_states(0, recs)
    _same_states(g, o, range(n))
    cfg = gw.config_from_json(_maze_config(13, 9, 5))
    for e in (g, o):
        e.set_config(cfg)
        e.new_game()
    for t in range(200):
        a = synthetic_actions("gridworld", n, t, seed=2)
        rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)
        for x, y in zip(rg, ro):
            assert np.array_equal(x, y)
    _same_states(g, o, range(n))
    assert np.array_equal(g.render(1), o.render(1))


@pytest.mark.gpu
def test_gpu_agent_pipeline(hip_lib, oracle_lib):
    n = 256
    g, o = _pair(n, hip_lib, oracle_lib)
    for e in (g, o):
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=True, fire_reset=True, noop_max=10,
                     noop_seed=3)
    assert np.array_equal(g.agent_reset(), o.agent_reset())
    ends = 0
    for t in range(400):
        a = synthetic_actions("gridworld", n, t, seed=6)
        x, y = g.agent_step(a), o.agent_step(a)
        for p, q in zip(x, y):
            assert np.array_equal(p, q), t
        eg, eo = g.agent_episodes(), o.agent_episodes()
        assert np.array_equal(eg[0], eo[0]) and np.array_equal(eg[1][eg[0]], eo[1][eo[0]])
        ends += int(eg[0].sum())
    assert ends > 0
    _same_states(g, o, range(0, n, 3))
