"""Edge cases of the boundary, run against the CPU restatement of the ABI here (host logic) and against the HIP library
on the GPU box (same assertions): ragged batch sizes, masks, single-env calls, error codes, capacity limits."""
import numpy as np
import pytest

from conftest import has_gpu
from support import synthetic_actions
from toybox_amd import Engine, ToyboxAmdError, _abi
from toybox_amd.games import codec

GAMES = ["breakout", "space_invaders", "amidar"]


def _libs(request, hip_lib, oracle_lib):
    return hip_lib if request.param == "hip" else oracle_lib


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request, oracle_lib):
    if request.param == "oracle":
        return oracle_lib
    from toybox_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("game", GAMES + ["gridworld"])
@pytest.mark.parametrize("n", [1, 3, 5, 67])
def test_ragged_batch_sizes(game, n, lib, oracle_lib):
    """Batch sizes that do not fill a 4-env block / a 64-lane wave behave like any other: every env steps, renders and
    round-trips, and equals the oracle (trivially so when lib IS the oracle)."""
    e, o = Engine(game, n, lib=lib), Engine(game, n, lib=oracle_lib)
    for x in (e, o):
        x.seed(77)
        x.new_game()
    for t in range(120 if game != "gridworld" else 900):     # long enough for GridWorld games to end (auto-reset in a partly empty wave)
        a = synthetic_actions(game, n, t)
        r1, r2 = e.step(a, auto_reset=True), o.step(a, auto_reset=True)
        for p, q in zip(r1, r2):
            assert np.array_equal(p, q)
    assert np.array_equal(e.render(3), o.render(3))
    assert np.array_equal(e.render_env(n - 1, 1), o.render_env(n - 1, 1))
    for i in range(n):
        assert bytes(e.get_state(i)) == bytes(o.get_state(i))


@pytest.mark.parametrize("game", GAMES)
def test_masked_new_game_and_single_env_input(game, lib):
    n = 6
    e = Engine(game, n, lib=lib)
    e.seed(5)
    e.new_game()
    for t in range(40):
        e.step(synthetic_actions(game, n, t))
    before = [bytes(e.get_state(i)) for i in range(n)]
    mask = np.array([0, 1, 0, 0, 1, 0], np.uint8)
    e.new_game(mask)
    cd = codec(game)
    for i in range(n):
        if mask[i]:
            js = cd.state_to_json(e.get_state(i))
            assert js["score"] == 0 and bytes(e.get_state(i)) != before[i]
        else:
            assert bytes(e.get_state(i)) == before[i], "unmasked env %d was touched" % i
    e.apply_input(2, _abi.BTN_BUTTON1)
    for i in (0, 3, 5):
        assert bytes(e.get_state(i)) == before[i]
    assert bytes(e.get_state(2)) != before[2]


def test_error_codes(lib):
    with pytest.raises(ToyboxAmdError) as ei:
        Engine("breakout", 0, lib=lib)
    assert ei.value.code == _abi.E_INVALID
    e = Engine("breakout", 2, lib=lib)
    for bad_call in (lambda: e.get_state(2), lambda: e.get_state(-1), lambda: e.render_env(5, 3),
                     lambda: e.render(2), lambda: e.apply_input(9, 0), lambda: e.seed(1, env=7),
                     lambda: e.query(0, 1, [1, 2])):
        with pytest.raises(ToyboxAmdError) as ei:
            bad_call()
        assert ei.value.code == _abi.E_INVALID
    with pytest.raises(ValueError):
        e.step([0])                      # wrong batch shape is caught on the host
    e.step([0, 1])                       # the engine is still usable afterwards


def test_capacity_limits_are_reported(lib):
    """States outside what the device engine holds are refused loudly (TBX_E_UNSUPPORTED), never truncated."""
    e = Engine("breakout", 1, lib=lib)
    st = e.get_state(0)
    st.n_balls = 5
    with pytest.raises(ToyboxAmdError) as ei:
        e.set_state(0, st)
    assert ei.value.code == _abi.E_UNSUPPORTED
    st.n_balls = 4                       # the maximum is fine
    for b in range(4):
        st.ball_x[b], st.ball_y[b], st.ball_vx[b], st.ball_vy[b] = 60.0 + 20 * b, 100.0, 1.0, -1.5
    st.is_dead = 0
    e.set_state(0, st)
    for t in range(300):
        e.step([3 + t % 2])
    assert e.get_state(0).n_balls <= 4
    cfg = e.get_config()
    cfg.paddle_discrete_segments = 0
    with pytest.raises(ToyboxAmdError) as ei:
        e.set_config(cfg)
    assert ei.value.code == _abi.E_UNSUPPORTED
    s = Engine("space_invaders", 1, lib=lib)
    st = s.get_state(0)
    st.n_enemy_lasers = 9
    with pytest.raises(ToyboxAmdError) as ei:
        s.set_state(0, st)
    assert ei.value.code == _abi.E_UNSUPPORTED
    st = s.get_state(0)
    for field, bad in (("row", 256), ("col", -1), ("id", 65536)):     # the enemy table packs these three into one word
        st2 = type(st).from_buffer_copy(bytes(st))
        setattr(st2.enemies[3], field, bad)
        with pytest.raises(ToyboxAmdError) as ei:
            s.set_state(0, st2)
        assert ei.value.code == _abi.E_UNSUPPORTED
    st.enemies[3].row, st.enemies[3].col, st.enemies[3].id = 255, 255, 65535      # the extremes round-trip
    s.set_state(0, st)
    back = s.get_state(0)
    assert (back.enemies[3].row, back.enemies[3].col, back.enemies[3].id) == (255, 255, 65535)
    a = Engine("amidar", 1, lib=lib)
    st = a.get_state(0)
    st.n_enemies = 9
    with pytest.raises(ToyboxAmdError) as ei:
        a.set_state(0, st)
    assert ei.value.code == _abi.E_UNSUPPORTED


@pytest.mark.parametrize("game", GAMES)
def test_seed_semantics(game, lib):
    """set_seed takes effect at the next new_game (envs/atari/base.py:95-97); equal seeds give equal games, env i of a
    batch seeded with s equals a single env seeded with s+i."""
    e = Engine(game, 3, lib=lib)
    before = bytes(e.get_state(1))
    e.seed(900)
    assert bytes(e.get_state(1)) == before           # nothing changes until the new game
    e.new_game()
    single = Engine(game, 1, lib=lib)
    single.seed(901)
    single.new_game()
    assert bytes(single.get_state(0)) == bytes(e.get_state(1))
    acts = [synthetic_actions(game, 3, t) for t in range(200)]
    for a in acts:
        e.step(a)
        single.step(a[1:2])
    assert bytes(single.get_state(0)) == bytes(e.get_state(1))


@pytest.mark.parametrize("game", GAMES + ["gridworld"])
def test_set_state_stores_the_canonical_record(game, lib, oracle_lib):
    """Hand-written records with junk in unused slots, flags that are not 0 / 1 and wide direction fields come back from
    tbx_get_state in ONE canonical form on both libraries, and writing that form back changes nothing."""
    e, o = Engine(game, 2, lib=lib), Engine(game, 2, lib=oracle_lib)
    st = o.get_state(0)
    raw = bytearray(bytes(st))
    if game == "breakout":
        st.is_dead, st.reset = 7, 200
        st.n_balls = 1
        st.ball_x[3], st.ball_vy[2] = 1.5e9, -3.25
        st.bricks[5].alive, st.bricks[6].destructible = 9, 77
        st.bricks[200].x, st.bricks[200].alive = 123.0, 1          # beyond n_bricks
    elif game == "space_invaders":
        st.has_ship_laser, st.ship_alive, st.ship_death_hit_1, st.visual_orientation = 5, 3, 4, 9
        st.ship_death_counter, st.ufo_death_counter = -7, -2
        st.move_dir = 7
        st.n_enemies = 30
        st.enemies[40].x, st.enemies[40].alive = 999, 1               # beyond n_enemies
        st.enemies[3].alive, st.enemies[3].death_counter = 8, -5
        st.n_enemy_lasers = 1
        st.enemy_lasers[0].movement = 6
        st.enemy_lasers[5].x = 44
        st.ship_laser.y = 55
        st.has_ship_laser = 0
        st.n_shields = 2
        st.shield_rows[2][4] = 0xFFFF
    elif game == "amidar":
        st.n_enemies = 3
        st.enemies[6].x, st.enemies[6].ai.kind = 5, 2
        st.n_boxes = 20
        st.boxes[40].tl_tx, st.boxes[40].painted = 9, 1
        st.boxes[2].painted, st.boxes[2].triggers_chase = 6, 4
        st.tiles[3][4] = 2 + 4 * 13                                     # only the low two bits are the tag
    else:
        st.game_over = 5
        st.tiles[0].goal, st.tiles[1].walkable = 3, 200
    assert bytes(st) != bytes(raw)
    for x in (e, o):
        x.set_state(1, st)
    a, b = e.get_state(1), o.get_state(1)
    assert bytes(a) == bytes(b)
    assert bytes(a) != bytes(st)                                        # it was normalised
    for x in (e, o):
        x.set_state(0, a)
    assert bytes(e.get_state(0)) == bytes(a) == bytes(o.get_state(0))   # the canonical form is a fixed point
    for t in range(30):
        act = synthetic_actions(game, 2, t)
        for p, q in zip(e.step(act), o.step(act)):
            assert np.array_equal(p, q)
    for i in range(2):
        assert bytes(e.get_state(i)) == bytes(o.get_state(i))
