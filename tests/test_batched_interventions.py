"""Batched interventions (8f rank 3): vectorised state get/set over env ranges on both libraries; the batched path must
equal env-by-env get_state/set_state, and engines driven through it must stay bit-identical to the oracle."""
import numpy as np
import pytest

from support import synthetic_actions
from toybox_amd import Engine
from toybox_amd.interventions import BatchIntervention


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request, oracle_lib):
    if request.param == "oracle":
        return oracle_lib
    from toybox_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("game", ["breakout", "space_invaders", "amidar"])
def test_batched_get_set_equals_single(game, lib):
    n = 37
    e = Engine(game, n, lib=lib)
    e.seed(3)
    e.new_game()
    for t in range(60):
        e.step(synthetic_actions(game, n, t))
    arr = e.get_states(5, 20)
    for k in range(20):
        assert bytes(arr[k]) == bytes(e.get_state(5 + k))
    rec = e.get_states_np()
    assert rec.shape == (n,) and int(rec["lives"][0]) == e.get_state(0).lives
    # write a shifted copy back: env i takes the state of env i+1
    e.set_states(0, e.get_states(1, n - 1))
    for k in range(n - 1):
        assert bytes(e.get_state(k)) == bytes(rec[k + 1].tobytes())


def test_breakout_channel_sweep(lib, oracle_lib):
    """add_channel in 64 envs with one vectorised write (the reference does this one env and one JSON round-trip at a
    time, test_breakout_interventions.py:60-76), then play on both engines."""
    n = 64
    e, o = Engine("breakout", n, lib=lib), Engine("breakout", n, lib=oracle_lib)
    for x in (e, o):
        x.seed(11)
        x.new_game()
        with BatchIntervention(x) as bi:
            before = bi.breakout_bricks_remaining()
            bi.breakout_add_channel(np.arange(n) % 18 if False else 4)
            bi.states["lives"] = 2
            assert bi.dirty_state
            assert (bi.breakout_bricks_remaining() == before - 6).all()
        with BatchIntervention(x) as bi:
            assert not bi.dirty_state and (bi.states["lives"] == 2).all()
            js = bi.json(7)
            assert sum(1 for b in js["bricks"] if b["col"] == 4 and b["alive"]) == 0
    for t in range(400):
        a = synthetic_actions("breakout", n, t)
        r1, r2 = e.step(a, auto_reset=True), o.step(a, auto_reset=True)
        for p, q in zip(r1, r2):
            assert np.array_equal(p, q)
    assert bytes(e.get_states()) == bytes(o.get_states())


def test_mode_sweeps(lib):
    a = Engine("amidar", 9, lib=lib)
    with BatchIntervention(a, 2, 5) as bi:
        bi.amidar_set_mode("jump")
    jt = a.get_states_np()["jump_timer"]
    assert (jt[2:7] == 75).all() and (jt[:2] == 0).all() and (jt[7:] == 0).all()
    # per-env times: a zero means the config's default for THAT env, like the reference's `set_time or config[...]`
    with BatchIntervention(a, 2, 5) as bi:
        bi.set_mode("chase", set_time=np.array([0, 11, 0, 12, 13]))
        bi.set_mode("jump", set_time=np.zeros(5, int), envs=[0, 1])
    st = a.get_states_np()
    assert list(st["chase_timer"][2:7]) == [300, 11, 300, 12, 13] and list(st["jump_timer"][2:4]) == [75, 75]
    # a `.states` array taken before a device-side helper is stale after it: writing through it raises instead of being lost
    with BatchIntervention(a) as bi:
        old = bi.states
        old["lives"][0] = 2
        bi.set_jumps(1)                                      # flushes the edit above, drops the host copy
        with pytest.raises(ValueError):
            old["lives"][1] = 1
        assert bi.states["lives"][0] == 2 and (bi.states["jumps"] == 1).all()
        bi.states["lives"][1] = 1                            # the fresh array takes edits
    assert list(a.get_states_np()["lives"][:2]) == [2, 1]
    s = Engine("space_invaders", 4, lib=lib)
    with BatchIntervention(s) as bi:
        bi.space_invaders_remove_mothership()
    for t in range(700):
        s.step([0] * 4)
    assert (s.get_states_np()["ufo_appearance_counter"] == -1).all()
    with pytest.raises(Exception):
        s.get_states(2, 5)


def test_set_eq_diff_and_partial_config(oracle_lib, tmp_path):
    """SetEq for batches + set_partial_config (interventions/base.py:47-106, 409-419)."""
    import json
    from toybox_amd.interventions import diff_states
    with Engine("breakout", 6, lib=oracle_lib) as e:
        e.seed(3)
        e.new_game()
        base = e.get_states_np()
        with BatchIntervention(e) as bi:
            assert bi.differs(base) == []
            bi.states["lives"][2] = 1
            bi.states["bricks"]["alive"][4, 17] = 0
            bi.states["paddle_x"][5] += 1e-13           # below the isclose tolerance: not a difference
            bi.states["paddle_x"][1] += 0.5
            d = dict((k, list(v)) for k, v in bi.differs(base))
            assert d == {"lives": [2], "bricks.alive": [4], "paddle_x": [1]}
        after = e.get_states_np()
        assert [k for k, _ in diff_states(after, base, rel_tol=0.0)] == ["lives", "paddle_x", "bricks.alive"] or \
            sorted(k for k, _ in diff_states(after, base, rel_tol=0.0)) == ["bricks.alive", "lives", "paddle_x"]
        # partial config from a file: unknown keys are ignored, known ones replace; the batch restarts with it
        f = tmp_path / "partial.json"
        f.write_text(json.dumps({"start_lives": 2, "no_such_key": 1}))
        with BatchIntervention(e) as bi:
            bi.set_partial_config(str(f))
            bi.set_partial_config({"ball_speed_slow": 1.5})
            assert bi.dirty_config
        cfg = e.get_config()
        assert cfg.start_lives == 2 and cfg.ball_speed_slow == 1.5
        assert (e.scalars()[1] == 2).all()


@pytest.mark.parametrize("game", ["breakout", "space_invaders", "amidar"])
def test_config_intervention_keeps_per_env_rngs(game, lib):
    """A batch-wide config edit (Intervention.set_partial_config + the new game of interventions/base.py:401-403) must not put
    every env on env 0's random stream: `rand` as tbx_get_config reported it means "not edited"; an edited `rand` is
    written to every env (the one-env case of both is the reference's behaviour)."""
    n = 9
    e = Engine(game, n, lib=lib)
    e.seed(500)
    e.new_game()
    before = [e.get_sim_rng(i) for i in range(n)]
    assert len(set(before)) == n
    key = {"breakout": "start_lives", "space_invaders": "start_lives", "amidar": "start_lives"}[game]
    with BatchIntervention(e) as bi:
        bi.set_partial_config({key: 2})
    assert (e.get_states_np()["lives"] == 2).all()
    # the new game drew one child from every env's own stream: all distinct, none equal to env 0's
    after = [e.get_sim_rng(i) for i in range(n)]
    assert len(set(after)) == n and all(a != b for a, b in zip(after, before))
    rngs = {tuple(int(v) for v in r) for r in e.get_states_np()["rand"]}
    assert len(rngs) == n, "envs share a state RNG after a config intervention"
    # an edited `rand` IS a request to re-seed every env
    with BatchIntervention(e) as bi:
        bi.config["rand"] = {"state": [123, 456]}
    assert len({e.get_sim_rng(i) for i in range(n)}) == 1


def test_seed_array_equals_per_env_seed(lib):
    n = 11
    a, b = Engine("space_invaders", n, lib=lib), Engine("space_invaders", n, lib=lib)
    seeds = [(977 * i * i + 13) % 2 ** 31 for i in range(n)]
    a.seed_array(seeds)
    for i, s in enumerate(seeds):
        b.seed(s, env=i)
    a.new_game()
    b.new_game()
    for i in range(n):
        assert a.get_sim_rng(i) == b.get_sim_rng(i) and bytes(a.get_state(i)) == bytes(b.get_state(i))
    with pytest.raises(ValueError):
        a.seed_array(seeds[:-1])


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's helper methods (interventions/breakout.py:303-429, amidar.py:360-615, space_invaders.py:165-176) as
# device-side kernels (tbx_edit / tbx_reduce).  Three checks: (1) every query against a second formulation -- numpy over the
# POD records -- on whichever library runs; (2) HIP against the oracle after a sequence of edits, states and rollout;
# (3) (build container only) the oracle's batched forms against the reference's OWN single-env classes driven through the
# ctoybox shim, tests/interventions_reference_worker.py.

def _played(game, n, lib, frames=260, seed=17):
    e = Engine(game, n, lib=lib)
    e.seed(seed)
    e.new_game()
    for t in range(frames):
        e.step(synthetic_actions(game, n, t, seed=3), auto_reset=True)
    return e


def test_breakout_helpers_against_numpy(lib):
    n = 41
    e = _played("breakout", n, lib)
    rng = np.random.default_rng(5)
    with BatchIntervention(e) as bi:
        st = e.get_states_np()
        live = (np.arange(256)[None, :] < st["n_bricks"][:, None]) & (st["bricks"]["alive"] != 0)
        assert np.array_equal(bi.num_bricks_remaining(), live.sum(axis=1))
        assert np.array_equal(bi.num_bricks(), st["n_bricks"]) and bi.num_rows() == 6 and (bi.num_columns() == 18).all()
        cols = rng.integers(0, 18, n)
        bi.add_channel(cols)                                      # a different column in every env
        bi.add_channel(3, envs=np.arange(n) % 2 == 0)             # and column 3 in the even ones
        st = e.get_states_np()
        for i in range(n):
            b = st["bricks"][i][:st["n_bricks"][i]]
            assert not b["alive"][b["col"] == cols[i]].any()
            assert (not b["alive"][b["col"] == 3].any()) if i % 2 == 0 or cols[i] == 3 else True
        chan = np.array([[not st["bricks"][i]["alive"][:108][st["bricks"][i]["col"][:108] == c].any() for c in range(18)] for i in range(n)])
        assert np.array_equal(bi.channel_count(), chan.sum(axis=1))
        assert np.array_equal(bi.find_channel(), np.where(chan.any(axis=1), chan.argmax(axis=1), -1))
        assert np.array_equal(bi.is_channel(3), chan[:, 3]) and np.array_equal(bi.is_channel(cols), chan[np.arange(n), cols])
        col7 = bi.get_column(7)
        assert col7.shape == (n, 6)
        for i in range(n):
            assert np.array_equal(col7[i], st["bricks"][i]["alive"][42:48])
        row2 = bi.get_row(2)
        assert row2.shape == (n, 18) and np.array_equal(row2, st["bricks"]["alive"][:, 2:108:6])
        # find_brick (:400-404): a host-evaluated predicate over the static attributes (or a mask) + the per-env alive flag
        bk = st["bricks"][:, :108]
        def first(sel):                                              # first True per env, -1 without one
            return np.where(sel.any(axis=1), sel.argmax(axis=1), -1)
        static = (bk["row"][0] >= 1) & (bk["col"][0] % 4 == 3)
        assert np.array_equal(bi.find_brick(lambda b: b.row >= 1 and b.col % 4 == 3, alive=True), first(static[None, :] & (bk["alive"] != 0)))
        assert np.array_equal(bi.find_brick(lambda b: b.row >= 1 and b.col % 4 == 3, alive=False), first(static[None, :] & (bk["alive"] == 0)))
        assert np.array_equal(bi.find_brick(static), np.full(n, int(static.argmax())))
        assert (bi.find_brick(lambda b: b.color.r == 1 and b.color.g == 2) == -1).all() and (bi.find_brick(np.zeros(108, bool)) == -1).all()
        assert bi.brick_table()[7].points == int(bk["points"][0][7]) and bi.brick_table()[7].position.x == float(bk["x"][0][7])
        bi.fill_column(cols)
        assert not bi.is_channel(cols).any()
        bi.clear_board(envs=[1, 5])
        rem = bi.num_bricks_remaining()
        assert rem[1] == 0 and rem[5] == 0 and (rem[[0, 2, 3]] > 0).all()
        bi.set_row_alive(0, 1)
        bi.set_brick_alive(107, 0)
        assert (bi.get_row(0) == 1).all() and (bi.get_column(17)[:, 5] == 0).all()
        pp = bi.get_paddle_position()
        assert np.array_equal(pp[:, 0], st["paddle_x"]) and np.array_equal(pp[:, 1], st["paddle_y"])
        assert np.array_equal(bi.get_paddle_velocity()[:, 0], st["paddle_vx"])
        nb, pos = bi.get_ball_position()
        assert np.array_equal(nb, st["n_balls"]) and np.array_equal(pos[:, 0, 0], np.where(nb > 0, st["ball_x"][:, 0], -1.0))
        bi.set_paddle_position(100.5 + np.arange(n) * 0.25)
        bi.set_ball(0, 60.0, 90.0, 1.0, -1.5, envs=nb > 0)
        bi.set_lives(np.arange(n) % 4 + 1)
        st2 = e.get_states_np()
        assert np.array_equal(st2["paddle_x"], 100.5 + np.arange(n) * 0.25) and np.array_equal(st2["lives"], np.arange(n) % 4 + 1)
        assert (st2["ball_vy"][nb > 0, 0] == -1.5).all() and np.array_equal(st2["ball_x"][nb == 0], st["ball_x"][nb == 0])
        assert not bi.dirty_state and not bi.dirty_config      # device-side edits need no write-back
    e.close()


def test_amidar_helpers_against_numpy(lib):
    n = 23
    e = _played("amidar", n, lib, frames=400)
    with BatchIntervention(e, 2, n - 4) as bi:                    # a sub-range of the batch
        m = n - 4
        st = e.get_states_np()[2:n - 2]
        assert np.array_equal(bi.get_jump_mode(), st["jump_timer"] > 0) and np.array_equal(bi.get_chase_mode(), st["chase_timer"] > 0)
        assert np.array_equal(bi.get_regular_mode(), (st["jump_timer"] == 0) & (st["chase_timer"] == 0))
        ne = st["n_enemies"]
        caught = np.array([st["enemies"][i]["caught"][:ne[i]].any() for i in range(m)])
        assert np.array_equal(bi.any_enemy_caught(), caught)
        tiles = st["tiles"]
        for tx, ty in ((0, 0), (31, 30), (5, 6), (1, 1)):
            assert list(bi.get_tile_by_pos(tx, ty)) == [["Empty", "Unpainted", "Painted", "ChaseMarker"][v] for v in tiles[:, ty, tx]]
            assert np.array_equal(bi.is_tile_walkable(tx, ty), tiles[:, ty, tx] != 0)
        for k, tag in enumerate(["Empty", "Unpainted", "Painted", "ChaseMarker"]):
            assert np.array_equal(bi.count_tiles(tag), (tiles == k).reshape(m, -1).sum(axis=1))
        adj = bi.get_adjacent_tiles(6, 6)
        assert np.array_equal(adj, np.stack([tiles[:, 5, 6], tiles[:, 6, 5], tiles[:, 6, 7], tiles[:, 7, 6]], axis=1))
        assert (bi.get_adjacent_tiles(0, 0)[:, :2] == -1).all()
        ptx, pty = st["player"]["x"] // 64, st["player"]["y"] // 80
        gx, gy, gtag = bi.player_tile()
        assert np.array_equal(gx, ptx) and np.array_equal(gy, pty)
        etx, ety = st["enemies"]["x"] // 64, st["enemies"]["y"] // 80
        want = np.abs(etx - ptx[:, None]) + np.abs(ety - pty[:, None])
        want[np.arange(8)[None, :] >= ne[:, None]] = -1
        assert np.array_equal(bi.player_enemy_distances(), want)
        want9 = np.abs(etx - 9) + np.abs(ety - 12)
        want9[np.arange(8)[None, :] >= ne[:, None]] = -1
        assert np.array_equal(bi.enemy_distances_from_tile(9, 12), want9)
        on = tiles[np.arange(m), pty, ptx] == 2
        assert np.array_equal(bi.player_on_painted(), on)
        yy, xx = np.mgrid[0:31, 0:32]
        for radius in (1, 3, 5):
            near = (np.abs(xx[None] - ptx[:, None, None]) + np.abs(yy[None] - pty[:, None, None]) < radius) & (tiles != 0)
            assert np.array_equal(bi.player_near_unpainted(radius), (near & (tiles == 2)).reshape(m, -1).sum(1) != near.reshape(m, -1).sum(1))
        # mask / counter-RNG forms against numpy over the state records
        from support import splitmix64
        walk = tiles != 0
        assert np.array_equal(bi.filter_tiles(lambda tag: tag != "Empty"), walk) and np.array_equal(bi.filter_tiles("ChaseMarker"), tiles == 3)
        assert np.array_equal(bi.count_filtered_tiles(["Painted", "Unpainted"]), ((tiles == 1) | (tiles == 2)).reshape(m, -1).sum(1))
        rtx, rty, rtag, rcnt = bi.get_random_tile(lambda tag: tag != "Empty", seed=21, draw=5, env_offset=40)
        dirs = bi.get_random_dir_for_tile(rtx, rty, seed=21, draw=6, env_offset=40)
        far = bi.get_random_tile(seed=3, draw=0, min_enemy_distance=12)
        for i in range(m):
            cand = np.argwhere(walk[i])                            # row by row: (ty, tx)
            r = int(splitmix64(21 ^ ((40 + 2 + i) << 32) ^ 5))    # env index within the ENGINE (the range starts at env 2)
            assert rcnt[i] == len(cand) and (rty[i], rtx[i]) == tuple(cand[r % len(cand)]) and rtag[i] == TILE_NAMES[tiles[i, rty[i], rtx[i]]]
            ok = [name for name, (dx, dy) in (("Up", (0, -1)), ("Down", (0, 1)), ("Left", (-1, 0)), ("Right", (1, 0)))
                  if 0 <= rtx[i] + dx < 32 and 0 <= rty[i] + dy < 31 and tiles[i, rty[i] + dy, rtx[i] + dx] != 0]
            assert dirs[i] == ok[int(splitmix64(21 ^ ((40 + 2 + i) << 32) ^ 6)) % len(ok)]
            d = np.abs(etx[i, :ne[i], None, None] - xx[None]) + np.abs(ety[i, :ne[i], None, None] - yy[None])
            cand = np.argwhere(~(d < 12).all(axis=0))
            assert far[3][i] == len(cand) and (far[1][i], far[0][i]) == tuple(cand[int(splitmix64(3 ^ ((2 + i) << 32) ^ 0)) % len(cand)])
        none = bi.get_random_tile("Painted", seed=1, min_enemy_distance=200)       # nobody is 200 tiles away: no candidate anywhere
        assert (none[0] == -1).all() and (none[3] == 0).all() and all(t is None for t in none[2])
        before_pos = (st["player"]["x"].copy(), st["player"]["y"].copy())
        bi.set_player_random_start(12, seed=3, draw=0, envs=np.arange(m) % 2 == 0)
        after = e.get_states_np()[2:n - 2]["player"]
        for i in range(m):
            want_xy = (far[0][i] * 64, far[1][i] * 80) if i % 2 == 0 else (before_pos[0][i], before_pos[1][i])
            assert (after["x"][i], after["y"][i]) == want_xy
        bi.set_player_tile(ptx, pty)                                  # back, for the checks below
        # edits
        bi.set_mode("jump", envs=np.arange(m) < 5)
        bi.set_mode("chase", set_time=np.arange(m) + 7)
        bi.set_tile_tag(5, 6, "Painted")
        bi.set_tile_tag(np.arange(m) % 32, 0, "ChaseMarker", envs=np.arange(m) % 3 == 0)
        bi.set_jumps(5)
        bi.set_player_tile(31, 15, envs=[0])
        bi.set_enemy_protocol(1, "EnemyTargetPlayer", start={"tx": 0, "ty": 30}, start_dir="Right", vision_distance=9, dir="Up")
        bi.set_enemy_protocol(0, "EnemyPerimeterAI", start={"tx": 0, "ty": 0}, envs=[2, 3])
        all_st = e.get_states_np()
        s2 = all_st[2:n - 2]
        assert np.array_equal(s2["jump_timer"][:5], [75] * 5) and np.array_equal(s2["chase_timer"], np.arange(m) + 7)
        assert (s2["tiles"][:, 6, 5] == 2).all() and (s2["jumps"] == 5).all()
        for i in range(m):
            assert s2["tiles"][i, 0, i % 32] == (3 if i % 3 == 0 else tiles[i, 0, i % 32])
        assert s2["player"]["x"][0] == 31 * 64 and s2["player"]["y"][0] == 15 * 80
        ai = s2["enemies"]["ai"][:, 1]
        assert (ai["kind"] == 4).all() and (ai["vision_distance"] == 9).all() and (ai["dir"] == 0).all() and (ai["start_ty"] == 30).all()
        assert (s2["enemies"]["ai"]["kind"][[2, 3], 0] == 2).all()
        # the two envs on either side of the range were left alone
        for k in (0, 1, n - 2, n - 1):
            assert all_st["jumps"][k] != 5 or all_st["chase_timer"][k] == 0
    for t in range(120):                                          # and the game goes on from there
        e.step(synthetic_actions("amidar", n, t, seed=8), auto_reset=True)
    e.close()


def test_space_invaders_helpers(lib):
    n = 12
    e = _played("space_invaders", n, lib, frames=200)
    with BatchIntervention(e) as bi:
        st = e.get_states_np()
        ship = bi.get_player()
        assert np.array_equal(ship["x"], st["ship_x"]) and np.array_equal(ship["alive"], st["ship_alive"] != 0)
        assert np.array_equal(ship["death_counter"], st["ship_death_counter"]) and np.array_equal(ship["speed"], st["ship_speed"])
        bi.remove_mothership(envs=np.arange(n) < 6)
        bi.set_lives(1, envs=[n - 1])
        assert bi.get_jitter() == 0.5
    s2 = e.get_states_np()
    assert (s2["ufo_appearance_counter"][:6] == -1).all() and (s2["ufo_appearance_counter"][6:] >= 0).all() and s2["lives"][n - 1] == 1
    with BatchIntervention(e) as bi:
        bi.set_jitter(0.25)                                       # config intervention: new game on exit
    assert e.get_config().jitter == 0.25 and (e.scalars()[0] == 0).all()
    with pytest.raises(TypeError):
        with BatchIntervention(e) as bi:
            bi.channel_count()
    with pytest.raises(Exception):
        e.reduce(_abi_mod().QUERY_BRK_BALLS)
    e.close()


def _abi_mod():
    from toybox_amd import _abi
    return _abi


TILE_NAMES = ["Empty", "Unpainted", "Painted", "ChaseMarker"]


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["breakout", "amidar", "space_invaders"])
def test_device_side_interventions_hip_equals_oracle(game, oracle_lib):
    """The same sequence of batched helper calls on the HIP library and on the oracle (4 096 envs, per-env arguments and
    masks): every query answer, every state record of a sample and the rollout from there are identical.  Breakout also with
    intervention-written (non-canonical) bricks, where column / row come from the per-env tables."""
    from toybox_amd import _lib
    n = 4096
    hip_lib = _lib.load()
    g, o = _played(game, n, hip_lib, frames=300), _played(game, n, oracle_lib, frames=300)
    rng = np.random.default_rng(11)
    cols, mask = rng.integers(0, 18, n), rng.random(n) < 0.4
    tx, ty = rng.integers(0, 32, n), rng.integers(0, 31, n)
    answers = []
    for e in (g, o):
        out = []
        with BatchIntervention(e) as bi:
            if game == "breakout":
                for rnd in range(2):
                    bi.add_channel(cols, envs=mask)
                    bi.fill_column((cols + 5) % 18)
                    bi.set_row_alive(3, 0, envs=~mask)
                    bi.set_paddle_position(40.0 + (np.arange(n) % 150))
                    bi.set_ball(0, 100.0, 100.0, 1.25, -1.75, envs=mask)
                    out += [bi.num_bricks_remaining(), bi.num_bricks(), bi.channel_count(), bi.find_channel(), bi.is_channel(cols),
                            bi.get_column(4), bi.get_row(1), bi.get_paddle_position(), bi.get_ball_position()[1], bi.get_ball_velocity()[0],
                            bi.find_brick(lambda b: b.row >= 2 and b.col % 3 == 1, alive=True), bi.find_brick(np.arange(108) > 50, alive=False),
                            bi.find_brick(lambda b: b.points == 7)]
                    if rnd == 0:                                   # now the same on per-env brick tables
                        js = bi.json(7)
                        js["bricks"][5]["col"] = 17
                        js["bricks"][20]["position"]["x"] += 1.0
                        bi.write_json(7, js)
            elif game == "amidar":
                bi.set_mode("jump", set_time=1 + (np.arange(n) % 50), envs=mask)
                bi.set_mode("chase", envs=~mask)
                bi.set_tile_tag(tx, ty, "Painted", envs=mask)
                bi.set_player_tile(tx, ty, envs=np.arange(n) % 97 == 0)
                bi.set_enemy_protocol(2, "EnemyRandomMvmt", start={"tx": 12, "ty": 12}, start_dir="Left", dir="Left", envs=mask)
                bi.set_enemy_protocol(0, "EnemyAmidarMvmt", vert="Down", horiz="Right", start_vert="Down", start_horiz="Right", start={"tx": 6, "ty": 0})
                bi.set_jumps(np.arange(n) % 6)
                bi.set_player_random_start(6, seed=9, draw=3, env_offset=123456, envs=np.arange(n) % 5 == 1)
                rt = bi.get_random_tile(lambda tag: tag != "Empty", seed=4, draw=np.arange(n) % 7, env_offset=99)
                out += [bi.filter_tiles("Unpainted"), bi.count_filtered_tiles(lambda tag: tag in ("Painted", "ChaseMarker")), rt[0], rt[1], rt[3],
                        np.stack(bi.get_random_tile(seed=1, min_enemy_distance=20)[:2]), np.stack(bi.get_random_track_position(seed=8, draw=2)),
                        bi.get_random_dir_for_tile(tx, ty, seed=2, draw=1) == "Up", bi.get_random_dir_for_tile(tx, ty, seed=2, draw=1) == None]  # noqa: E711
                out += [bi.get_jump_mode(), bi.get_chase_mode(), bi.any_enemy_caught(), bi.count_tiles("Painted"), bi.count_tiles("Empty"),
                        bi.get_adjacent_tiles(tx, ty), bi.enemy_distances_from_tile(tx, ty), np.stack(bi.player_tile()[:2]),
                        bi.player_enemy_distances(), bi.player_on_painted(), bi.player_near_unpainted(4), bi.is_tile_walkable(tx, ty)]
            else:
                bi.remove_mothership(envs=mask)
                bi.set_lives(1 + np.arange(n) % 3)
                out += list(bi.get_player().values())
        answers.append(out)
    for k, (x, y) in enumerate(zip(*answers)):
        assert np.array_equal(np.asarray(x), np.asarray(y)), (game, k)
    sample = list(range(0, n, 61)) + [7]
    for i in sample:
        assert bytes(g.get_state(i)) == bytes(o.get_state(i)), (game, i)
    for t in range(300):
        a = synthetic_actions(game, n, t, seed=21)
        for x, y in zip(g.step(a, auto_reset=True), o.step(a, auto_reset=True)):
            assert np.array_equal(x, y), (game, t)
    assert np.array_equal(g.render(3)[::97], o.render(3)[::97])
    for i in sample:
        assert bytes(g.get_state(i)) == bytes(o.get_state(i)), (game, i)


@pytest.mark.gpu
def test_device_pointer_forms_of_edit_and_reduce(oracle_lib):
    """tbx_edit_device / tbx_reduce_device: mask, per-env arguments and results all in HBM, queued on a caller's stream between
    rollout steps without a synchronisation -- same answers as the host-pointer forms, and the edits land in program order."""
    from toybox_amd import _abi, _lib, hip
    n = 5000
    g, o = _played("breakout", n, _lib.load(), frames=100), _played("breakout", n, oracle_lib, frames=100)
    rng = np.random.default_rng(4)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    cols = np.stack([rng.integers(0, 18, n).astype(np.float64), np.zeros(n)], axis=1)          # {col, alive = 0} per env
    s = hip.Stream()
    d_mask, d_args, d_out = hip.malloc(n), hip.malloc(8 * 2 * n), hip.malloc(8 * n)
    hip.memcpy_htod(d_mask, mask, n)
    hip.memcpy_htod(d_args, cols, cols.nbytes)
    for t in range(100, 110):
        g.step_synthetic(1337, t, auto_reset=True, stream=s.ptr)
    g.edit_device(_abi.EDIT_BRK_COLUMN_ALIVE, mask_ptr=d_mask, stream=s.ptr, per_env_ptr=d_args, n_args=2)
    g.edit_device(_abi.EDIT_SET_LIVES, [2], mask_ptr=d_mask, stream=s.ptr)
    g.reduce_device(_abi.QUERY_BRK_CHANNEL_COUNT, d_out, stream=s.ptr)
    for t in range(110, 120):
        g.step_synthetic(1337, t, auto_reset=True, stream=s.ptr)
    s.synchronize()
    got = np.empty(n, np.float64)
    hip.memcpy_dtoh(got, d_out, 8 * n)
    for t in range(100, 110):
        o.step(synthetic_actions("breakout", n, t, seed=1337), auto_reset=True)
    o.edit(_abi.EDIT_BRK_COLUMN_ALIVE, cols, mask)
    o.edit(_abi.EDIT_SET_LIVES, [2], mask)
    want = o.reduce(_abi.QUERY_BRK_CHANNEL_COUNT)[:, 0]
    for t in range(110, 120):
        o.step(synthetic_actions("breakout", n, t, seed=1337), auto_reset=True)
    assert np.array_equal(got, want) and want[mask != 0].min() >= 1
    assert g.reduce_width(_abi.QUERY_BRK_BALLS) == 17 and g.reduce_width(_abi.QUERY_AMI_MODE) < 0
    for i in range(0, n, 53):
        assert bytes(g.get_state(i)) == bytes(o.get_state(i)), i
    for p in (d_mask, d_args, d_out):
        hip.free(p)


def test_batched_helpers_equal_the_references_own_classes(oracle_lib):
    """Build container only: tests/interventions_reference_worker.py imports the reference's UNMODIFIED
    toybox.interventions.{breakout,amidar,space_invaders} (over the ctoybox shim, one-env oracle engines), replays each env's
    state into it, calls the reference's helper and compares with what the batched form answered for that env."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    ref = os.environ.get("TOYBOX_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "toybox", "interventions")):
        pytest.skip("the reference tree is only present in the build container")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "shim"), ROOT, os.path.join(ROOT, "tests"), ref]),
               PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "interventions_reference_worker.py")], cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "WORKER_OK" in p.stdout, (p.stdout + p.stderr)[-4000:]


def test_counter_rule_arguments_are_folded_or_refused(lib):
    """seeds above 32 bits used to saturate onto ONE stream on the device (TbxEditArgs::getu).  The Python layer folds a
    seed of any size to 32 bits (so two different 64-bit seeds name different streams) and refuses draws / offsets that do not fit."""
    from toybox_amd.interventions import BatchIntervention
    n = 12
    e = Engine("amidar", n, lib=lib)
    e.seed(5); e.new_game()
    big_a, big_b = 0xDEADBEEF12345678, 0xDEADBEEF12345679
    with BatchIntervention(e) as bi:
        a = np.stack(bi.get_random_tile(seed=big_a, draw=1)[:2])
        b = np.stack(bi.get_random_tile(seed=big_b, draw=1)[:2])
        a32 = np.stack(bi.get_random_tile(seed=(big_a & 0xFFFFFFFF) ^ (big_a >> 32), draw=1)[:2])
        assert np.array_equal(a, a32) and not np.array_equal(a, b)
        for bad in ({"draw": 1 << 32}, {"env_offset": -1}, {"seed": -3}, {"draw": np.array([0] * (n - 1) + [1 << 40])}):
            with pytest.raises(ValueError):
                bi.get_random_tile(**bad)
        with pytest.raises(ValueError):
            bi.set_player_random_start(seed=1, draw=1 << 33)
    e.close()
