"""The ctoybox.Toybox surface beyond what the reference's intervention tests reach
through the shim: frameskip, Input, frame formats, PNG export, JSON / config round trips, schemas, queries, the
Simulator / State views (call sites: envs/atari/base.py, scripts/utils/test_games.py:5-41, start_images_toybox:24-37)."""
import json
import struct
import zlib

import numpy as np
import pytest

from toybox_amd import Engine
from toybox_amd import toybox as tbm
from toybox_amd.toybox import Input, Simulator, State, Toybox, write_png

GAMES = ["breakout", "amidar", "space_invaders", "gridworld"]


@pytest.fixture(autouse=True, params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine_factory(request, oracle_lib):
    lib = oracle_lib if request.param == "oracle" else request.getfixturevalue("hip_lib")
    tbm.set_engine_factory(lambda game, n: Engine(game, n, lib=lib))
    yield
    tbm.set_engine_factory(None)


def read_png(path):
    """minimal reader for what write_png produces (8-bit, filter 0): returns an (H, W, C) array"""
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.append((tag, body))
        pos += 12 + n
    assert [t for t, _ in chunks][0] == b"IHDR" and chunks[-1][0] == b"IEND"
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    c = {0: 1, 2: 3, 6: 4}[ctype]
    data = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    rows = np.frombuffer(data, np.uint8).reshape(h, 1 + w * c)
    assert depth == 8 and (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, c)


def test_png_writer_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for c in (1, 3, 4):
        img = rng.integers(0, 256, (13, 17, c), dtype=np.uint8)
        p = str(tmp_path / ("t%d.png" % c))
        write_png(p, img)
        assert np.array_equal(read_png(p), img)
    write_png(str(tmp_path / "g.png"), img[:, :, 0])            # 2-D input = gray
    assert read_png(str(tmp_path / "g.png")).shape == (13, 17, 1)


@pytest.mark.parametrize("game", GAMES)
def test_frames_and_png_export(game, tmp_path):
    with Toybox(game, grayscale=True) as tb:
        h, w = tb.get_height(), tb.get_width()
        gray = tb.get_state()
        assert gray.shape == (h, w, 1) and gray.dtype == np.uint8
        rgb = tb.get_rgb_frame()
        assert rgb.shape == (h, w, 3)
        p = tmp_path / "frame.png"
        tb.save_frame_image(str(p))
        assert np.array_equal(read_png(str(p)), rgb)
        tb.save_frame_image(str(p).encode("utf-8"), grayscale=True)      # ALE passes bytes (MockALE.saveScreenPNG)
        assert np.array_equal(read_png(str(p)), gray)
    with Toybox(game, grayscale=False) as tb:
        rgba = tb.get_state()
        assert rgba.shape == (h, w, 4) and (rgba[..., 3] == 255).all() and np.array_equal(rgba[..., :3], rgb)


@pytest.mark.parametrize("game", GAMES)
def test_actions_frameskip_and_input(game):
    with Toybox(game) as a, Toybox(game, frameskip=3) as b:
        legal = a.get_legal_action_set()
        assert legal == sorted(legal) and 0 in legal
        with pytest.raises(ValueError):
            a.apply_ale_action(max(set(range(18)) - set(legal)))         # "Expected to apply action, but failed"
        for k in range(40):
            act = legal[k % len(legal)]
            for _ in range(4):
                a.apply_ale_action(act)                                  # frameskip + 1 frames per call
            b.apply_ale_action(act)
        assert a.state_to_json() == b.state_to_json()
        assert a.get_score() == b.get_score() and a.get_lives() == b.get_lives() and a.get_level() == b.get_level()
        assert a.game_over() == (a.get_lives() <= 0) or game == "gridworld"
    inp = Input()
    inp.set_input("left", "button1")
    assert (inp.left, inp.button1, inp.right) == (True, True, False) and inp.to_mask() == 1 | 16
    with pytest.raises(ValueError):
        inp.set_input("sideways")
    with Toybox(game) as tb:
        before = tb.state_to_json()
        with pytest.raises(TypeError):
            tb.apply_action("left")
        tb.apply_action(Input())                                         # a no-op frame is still a frame
        assert tb.state_to_json() == before or game != "gridworld"


@pytest.mark.parametrize("game", GAMES)
def test_json_round_trips_schema_and_views(game):
    with Toybox(game, seed=7) as tb:
        for _ in range(25):
            tb.apply_ale_action(tb.get_legal_action_set()[1])
        st, cfg = tb.state_to_json(), tb.config_to_json()
        assert tb.to_state_json() == st
        json.dumps(st), json.dumps(cfg)                                  # plain JSON types only
        schema = tb.schema_for_state()
        assert schema["type"] == "object" and set(schema["required"]) == set(st.keys())
        assert set(tb.schema_for_config()["required"]) <= set(cfg.keys())
        with Toybox(game, withstate=st) as clone:                        # ctor argument of ctoybox.Toybox
            assert clone.state_to_json() == st
            clone.apply_ale_action(0)
            tb.apply_ale_action(0)
            assert clone.state_to_json() == tb.state_to_json()
        tb.write_state_json(json.dumps(st))                              # strings are accepted like dicts
        assert tb.state_to_json() == st
        tb.write_config_json(cfg)                                        # restarts the game under the same config
        assert tb.get_score() == 0
        sim = Simulator(tb)
        assert sim.get_frame_width() == tb.get_width() and sim.get_frame_height() == tb.get_height()
        assert {k: v for k, v in sim.to_json().items() if k != "rand"} == {k: v for k, v in cfg.items() if k != "rand"}
        state = sim.new_game()
        assert isinstance(state, State) and state.score() == 0 and state.lives() == tb.get_lives() and not state.game_over()
        assert state.level() == tb.get_level() and bool(state)


def test_queries():
    with Toybox("breakout") as tb:
        assert tb.query_state_json("bricks_remaining") == 108 and tb.query_state_json("num_columns") == 18
        assert tb.query_state_json("num_rows") == 6 and tb.query_state_json("channels") == []
        assert tb.query_state_json("brick_live_by_index", "5") is True
        js = tb.state_to_json()
        for b in js["bricks"]:
            if b["col"] == 4:
                b["alive"] = False
        tb.write_state_json(js)
        assert tb.query_state_json("channels") == [4] and tb.query_state_json("count_channels") == 1
        assert tb.query_state_json("bricks_remaining") == 102
        with pytest.raises(ValueError):
            tb.query_state_json("no_such_query")
    with Toybox("amidar") as tb:
        assert tb.query_state_json("tile_to_world", {"tx": 3, "ty": 2}) == [192, 160]      # 64 x 80 world units per tile
        assert tb.query_state_json("world_to_tile", json.dumps({"x": 200, "y": 170})) == [3, 2]
        assert tb.query_state_json("jumps_remaining") == 4 and tb.query_state_json("num_tiles_unpainted") == 356
        assert State(tb).query_json("jumps_remaining") == 4
    with Toybox("space_invaders") as tb:
        assert tb.query_state_json("enemies_remaining") == 36 and tb.query_state_json("shield_count") == 3
        assert tb.query_state_json("ship_x") == 68


def test_unknown_game_and_shared_engine():
    with pytest.raises(ValueError):
        Toybox("pong")
    eng = tbm._make_engine("breakout", 3)
    views = [Toybox("breakout", engine=eng, env_index=i) for i in range(3)]
    views[1].apply_ale_action(1)
    views[1].apply_ale_action(3)
    assert views[0].state_to_json()["paddle"] == views[2].state_to_json()["paddle"] != views[1].state_to_json()["paddle"]
    views[1].new_game()
    assert views[1].state_to_json()["paddle"] == views[0].state_to_json()["paddle"]
    for v in views:
        v.close()
    eng.close()
