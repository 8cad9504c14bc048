"""Run by tests/test_batched_interventions.py in a subprocess, build container only (PYTHONPATH = tests/shim, the repo, the
reference tree).  The reference's OWN intervention classes -- toybox.interventions.{breakout,amidar,space_invaders}, imported
unmodified, over the `ctoybox` shim and one-env oracle engines -- are the yardstick for the batched helpers of
toybox_amd.interventions.BatchIntervention: every env's state of a played batch is written into a reference Toybox, the
reference's helper is called on it, and its answer (or the state it leaves behind) must equal what the batched form produced
for that env.  Builder-authored; contains no reference code."""
import json

import numpy as np

import ctoybox
from ctoybox import Toybox
from support import splitmix64, synthetic_actions
from toybox.interventions import amidar as ref_ami
from toybox.interventions.amidar import AmidarIntervention
from toybox.interventions.breakout import BreakoutIntervention
from toybox.interventions.core import Direction
from toybox.interventions.space_invaders import SpaceInvadersIntervention
from toybox_amd import Engine
from toybox_amd.games import codec
from toybox_amd.interventions import BatchIntervention

LIB = ctoybox._lib


def played(game, n, frames, seed):
    e = Engine(game, n, lib=LIB)
    e.seed(seed)
    e.new_game()
    for t in range(frames):
        e.step(synthetic_actions(game, n, t, seed=3), auto_reset=True)
    return e


def env_json(e, i):
    return codec(e.game).state_to_json(e.get_state(i))


def same_json(a, b, what):
    assert json.dumps(a, sort_keys=True) == json.dumps(b, sort_keys=True), what


def breakout():
    n = 10
    e = played("breakout", n, 300, 5)
    rng = np.random.default_rng(2)
    with BatchIntervention(e) as bi:                         # a few channels first, so that the counts are not all zero
        bi.add_channel(rng.integers(0, 18, n))
        bi.add_channel(2, envs=np.arange(n) % 2 == 0)
    before = [env_json(e, i) for i in range(n)]
    cols = rng.integers(0, 18, n)
    with BatchIntervention(e) as bi:
        q = dict(remaining=bi.num_bricks_remaining(), bricks=bi.num_bricks(), rows=bi.num_rows(), columns=bi.num_columns(),
                 count=bi.channel_count(), find=bi.find_channel(), col=bi.get_column(cols), is_chan=bi.is_channel(cols),
                 ppos=bi.get_paddle_position(), pvel=bi.get_paddle_velocity(), bpos=bi.get_ball_position(), bvel=bi.get_ball_velocity())
        # find_brick (:400-404): predicates over the static attributes, evaluated once on the host, + the per-env alive flag
        preds = {"row2_right": lambda b: b.row == 2 and b.col >= 5, "cheap": lambda b: b.points == 1, "red": lambda b: b.color.r > 150,
                 "none": lambda b: b.row > 99}
        q["find_alive"] = {k: bi.find_brick(p, alive=True) for k, p in preds.items()}
        q["find_dead"] = {k: bi.find_brick(p, alive=False) for k, p in preds.items()}
        q["find_any"] = {k: bi.find_brick(p) for k, p in preds.items()}
        q["find_mask"] = bi.find_brick(np.arange(108) % 7 == 3, alive=True)
        bi.add_channel(cols)
        bi.fill_column((cols + 1) % 18)
        bi.clear_board(envs=np.arange(n) == 4)

    def ref_find(iv, pred):
        try:
            return iv.find_brick(pred)[0]
        except ValueError:
            return -1
    for i in range(n):
        tb = Toybox("breakout")
        tb.write_state_json(before[i])
        with BreakoutIntervention(tb) as iv:
            for k, p in preds.items():
                assert ref_find(iv, lambda b, p=p: p(b) and b.alive) == q["find_alive"][k][i], (i, k)
                assert ref_find(iv, lambda b, p=p: p(b) and not b.alive) == q["find_dead"][k][i], (i, k)
                assert ref_find(iv, p) == q["find_any"][k][i], (i, k)
            idx = {id(b): j for j, b in enumerate(iv.game.bricks)}
            assert ref_find(iv, lambda b: idx[id(b)] % 7 == 3 and b.alive) == q["find_mask"][i]
            assert iv.num_bricks_remaining() == q["remaining"][i] and iv.num_bricks() == q["bricks"][i]
            assert iv.num_rows() == q["rows"] and iv.num_columns() == q["columns"][i]
            assert iv.channel_count() == q["count"][i], (i, iv.channel_count(), q["count"][i])
            assert iv.find_channel()[0] == q["find"][i]
            col = iv.get_column(int(cols[i]))
            assert [int(b.alive) for b in col] == list(q["col"][i]) and iv.is_stack(col)
            assert iv.is_channel(col) == bool(q["is_chan"][i])
            pp, pv = iv.get_paddle_position(), iv.get_paddle_velocity()
            assert (pp.x, pp.y) == tuple(q["ppos"][i]) and (pv.x, pv.y) == tuple(q["pvel"][i])
            nb = q["bpos"][0][i]
            if nb == 1:
                bp, bv = iv.get_ball_position(), iv.get_ball_velocity()
                assert (bp.x, bp.y) == tuple(q["bpos"][1][i, 0]) and (bv.x, bv.y) == tuple(q["bvel"][1][i, 0])
            iv.add_channel(int(cols[i]))
            iv.fill_column(int((cols[i] + 1) % 18))
            if i == 4:
                iv.clear_board()
        same_json(tb.to_state_json(), env_json(e, i), "breakout env %d after the edits" % i)
        tb.close() if hasattr(tb, "close") else None
    print("breakout ok")


def amidar():
    n = 8
    e = played("amidar", n, 500, 9)
    with BatchIntervention(e) as bi:
        bi.set_mode("chase", set_time=40, envs=[1, 2])       # so that modes differ over the batch
        bi.set_mode("jump", envs=[2, 3])
    before = [env_json(e, i) for i in range(n)]
    probe = [(0, 0), (5, 6), (31, 15), (12, 12), (1, 1)]
    tags = ["Empty", "Unpainted", "Painted", "ChaseMarker"]
    with BatchIntervention(e) as bi:
        q = dict(regular=bi.get_regular_mode(), jump=bi.get_jump_mode(), chase=bi.get_chase_mode(), caught=bi.any_enemy_caught(),
                 tile={p: bi.get_tile_by_pos(*p) for p in probe}, walk={p: bi.is_tile_walkable(*p) for p in probe},
                 count={t: bi.count_tiles(t) for t in tags}, dist={p: bi.enemy_distances_from_tile(*p) for p in probe},
                 ptile=bi.player_tile(), pdist=bi.player_enemy_distances(), painted=bi.player_on_painted(),
                 near={r: bi.player_near_unpainted(r) for r in (2, 5)})
        # the mask / counter-RNG forms of the predicate- and `random`-driven helpers
        walk_pred = lambda tag: tag != "Empty"
        q["filter"] = {"walk": bi.filter_tiles(walk_pred), "painted": bi.filter_tiles("Painted"), "all": bi.filter_tiles()}
        q["fcount"] = bi.count_filtered_tiles(["Painted", "ChaseMarker"])
        q["rtile"] = {k: bi.get_random_tile(walk_pred, seed=77, draw=k, env_offset=1000) for k in range(3)}
        q["rtile_far"] = bi.get_random_tile(seed=5, draw=1, min_enemy_distance=9)
        q["rtrack"] = bi.get_random_track_position(seed=77, draw=0, env_offset=1000)
        q["rdir"] = {p: bi.get_random_dir_for_tile(*p, seed=3, draw=2) for p in probe}
        bi.set_mode("regular", envs=[1])
        bi.set_mode("jump", set_time=33, envs=[0, 5])
        bi.set_mode("chase", envs=[6])
        bi.set_tile_tag(5, 6, "Painted")
        bi.set_tile_tag(1, 1, "Unpainted", envs=[7])
        bi.set_enemy_protocol(4, "EnemyPerimeterAI", start={"tx": 0, "ty": 0})
        bi.set_enemy_protocol(3, "EnemyAmidarMvmt", vert="Down", horiz="Left", start_vert="Up", start_horiz="Right", start={"tx": 6, "ty": 0}, envs=[0, 1, 2])
        bi.set_enemy_protocol(2, "EnemyTargetPlayer", start={"tx": 0, "ty": 30}, vision_distance=10, player_seen=None, start_dir="Right", dir="Up", envs=[3])
        bi.set_enemy_protocol(1, "EnemyRandomMvmt", start={"tx": 31, "ty": 30}, start_dir="Up", dir="Left", envs=[4])
        bi.set_enemy_protocol(0, "EnemyLookupAI", next=3, default_route_index=10, envs=[5])
    for i in range(n):
        tb = Toybox("amidar")
        tb.write_state_json(before[i])
        with AmidarIntervention(tb) as iv:
            assert iv.get_regular_mode() == q["regular"][i] and iv.get_jump_mode() == q["jump"][i] and iv.get_chase_mode() == q["chase"][i]
            assert bool(iv.any_enemy_caught(0)) == bool(q["caught"][i])
            for p in probe:
                t = iv.get_tile_by_pos(*p)
                assert t.tag == q["tile"][p][i] and iv.is_tile_walkable(t) == bool(q["walk"][p][i])
                assert iv.enemy_distances_from_tile(t) == [d for d in q["dist"][p][i] if d >= 0], (i, p)
            for t in tags:
                assert len(iv.filter_tiles(lambda x, t=t: x.tag == t)) == q["count"][t][i]
            ptp = iv.worldpoint_to_tilepoint(iv.game.player.position)
            assert (ptp.tx, ptp.ty) == (q["ptile"][0][i], q["ptile"][1][i]) and iv.player_tile().tag == q["ptile"][2][i]
            assert iv.player_enemy_distances() == [d for d in q["pdist"][i] if d >= 0]
            assert iv.player_on_painted() == bool(q["painted"][i])
            for r in (2, 5):
                assert iv.player_near_unpainted(r) == bool(q["near"][r][i]), (i, r)
            # filter_tiles(pred) in the reference's order; the drawn tile is element (r mod len) of that list
            def where(tiles):
                return [(iv.tile_to_tilepoint(t).tx, iv.tile_to_tilepoint(t).ty) for t in tiles]
            walk = where(iv.filter_tiles(lambda t: t.tag != "Empty"))
            assert walk == [(int(x), int(y)) for y, x in np.argwhere(q["filter"]["walk"][i])]
            assert len(iv.filter_tiles(lambda t: t.tag == "Painted")) == int(q["filter"]["painted"][i].sum())
            assert q["filter"]["all"][i].all() and len(iv.filter_tiles()) == q["filter"]["all"][i].size
            assert len(iv.filter_tiles(lambda t: t.tag in ("Painted", "ChaseMarker"))) == q["fcount"][i]
            for k in range(3):
                tx, ty, tag, cnt = (v[i] for v in q["rtile"][k])
                r = int(splitmix64(77 ^ ((1000 + i) << 32) ^ k))
                assert cnt == len(walk) and (tx, ty) == walk[r % len(walk)] and tag == iv.get_tile_by_pos(tx, ty).tag
                # the reference's own get_random_tile, its two randint draws scripted to land on that tile: accepted at once
                draws = iter([int(ty), int(tx)])
                ref_ami.random.randint = lambda a, b: next(draws)
                got = iv.get_random_tile(lambda t: t.tag != "Empty")
                assert got is iv.get_tile_by_pos(tx, ty)
            draws = iter([int(q["rtile"][0][1][i]), int(q["rtile"][0][0][i])])
            ref_ami.random.randint = lambda a, b: next(draws)
            wp = iv.get_random_track_position()
            assert (wp.x, wp.y) == (q["rtrack"][0][i], q["rtrack"][1][i])
            # set_player_random_start's predicate, as written: not all enemies nearer than the minimum
            def within(t, m=9):
                return not all(d < m for d in iv.enemy_distances_from_tile(t))
            far = where(iv.filter_tiles(within))
            tx, ty, _, cnt = (v[i] for v in q["rtile_far"])
            assert cnt == len(far) and (tx, ty) == far[int(splitmix64(5 ^ (i << 32) ^ 1)) % len(far)]
            # get_random_dir_for_tile cannot be called (it reads tile.tx, which the reference's Tile does not have): its rule --
            # a direction whose neighbour is walkable -- against is_tile_walkable of the reference
            for p in probe:
                ok = []
                for name, (dx, dy) in (("Up", (0, -1)), ("Down", (0, 1)), ("Left", (-1, 0)), ("Right", (1, 0))):
                    x, y = p[0] + dx, p[1] + dy
                    if 0 <= x < 32 and 0 <= y < 31 and iv.is_tile_walkable(iv.get_tile_by_pos(x, y)):
                        ok.append(name)
                want = ok[int(splitmix64(3 ^ (i << 32) ^ 2)) % len(ok)] if ok else None
                assert q["rdir"][p][i] == want, (i, p, ok)
            # the same edits through the reference's own methods
            if i == 1:
                iv.set_mode("regular")
            if i in (0, 5):
                iv.set_mode("jump", 33)
            if i == 6:
                iv.set_mode("chase")
            iv.set_tile_tag(iv.get_tile_by_pos(5, 6), "Painted")
            if i == 7:
                iv.set_tile_tag(iv.get_tile_by_pos(1, 1), "Unpainted")
            en = iv.game.enemies
            D = lambda name: Direction(iv, name)
            TP = lambda tx, ty: ref_ami.TilePoint(iv, tx=tx, ty=ty)
            iv.set_enemy_protocol(en[4], "EnemyPerimeterAI", start=TP(0, 0))
            if i in (0, 1, 2):
                iv.set_enemy_protocol(en[3], "EnemyAmidarMvmt", vert=D("Down"), horiz=D("Left"), start_vert=D("Up"), start_horiz=D("Right"), start=TP(6, 0))
            if i == 3:
                iv.set_enemy_protocol(en[2], "EnemyTargetPlayer", start=TP(0, 30), vision_distance=10, player_seen=None, start_dir=D("Right"), dir=D("Up"))
            if i == 4:
                iv.set_enemy_protocol(en[1], "EnemyRandomMvmt", start=TP(31, 30), start_dir=D("Up"), dir=D("Left"))
            if i == 5:
                iv.set_enemy_protocol(en[0], "EnemyLookupAI", next=3, default_route_index=10)
        got, want = tb.to_state_json(), env_json(e, i)
        got["board"]["junctions"], want["board"]["junctions"] = sorted(got["board"]["junctions"]), sorted(want["board"]["junctions"])
        same_json(got, want, "amidar env %d after the edits" % i)
    # set_player_random_start (:541-548): the batched edit, then the reference's method with its draws scripted to the tile the
    # counter rule chose for that env -- the reference must accept it at once and leave the same state
    before = [env_json(e, i) for i in range(n)]
    with BatchIntervention(e) as bi:
        chosen = bi.get_random_tile(seed=11, draw=4, env_offset=50, min_enemy_distance=5)
        bi.set_player_random_start(5, seed=11, draw=4, env_offset=50, envs=np.arange(n) != 2)
    for i in range(n):
        tb = Toybox("amidar")
        tb.write_state_json(before[i])
        if i != 2:
            with AmidarIntervention(tb) as iv:
                draws = iter([int(chosen[1][i]), int(chosen[0][i])])
                ref_ami.random.randint = lambda a, b: next(draws)
                iv.set_player_random_start(5)
        got, want = tb.to_state_json(), env_json(e, i)
        got["board"]["junctions"], want["board"]["junctions"] = sorted(got["board"]["junctions"]), sorted(want["board"]["junctions"])
        same_json(got, want, "amidar env %d after set_player_random_start" % i)
    print("amidar ok")


def space_invaders():
    n = 6
    e = played("space_invaders", n, 250, 4)
    before = [env_json(e, i) for i in range(n)]
    with BatchIntervention(e) as bi:
        ship = bi.get_player()
        assert bi.get_jitter() == 0.5
        bi.remove_mothership(envs=[0, 3])
        bi.set_lives(1, envs=[5])
    for i in range(n):
        tb = Toybox("space_invaders")
        tb.write_state_json(before[i])
        with SpaceInvadersIntervention(tb) as iv:
            s = iv.get_player()
            assert (s.x, s.y, s.w, s.h, s.speed, bool(s.alive)) == (ship["x"][i], ship["y"][i], ship["w"][i], ship["h"][i], ship["speed"][i], bool(ship["alive"][i]))
            assert (-1 if s.death_counter is None else s.death_counter) == ship["death_counter"][i]
            assert iv.get_jitter() == 0.5
            if i in (0, 3):
                iv.remove_mothership(None)
            if i == 5:
                iv.game.lives = 1
        same_json(tb.to_state_json(), env_json(e, i), "space_invaders env %d after the edits" % i)
    # set_jitter: a config intervention on both sides (new game on exit)
    with BatchIntervention(e) as bi:
        bi.set_jitter(0.125)
    tb = Toybox("space_invaders")
    with SpaceInvadersIntervention(tb) as iv:
        iv.set_jitter(0.125)
    assert tb.config_to_json()["jitter"] == 0.125 == codec("space_invaders").config_to_json(e.get_config())["jitter"]
    print("space_invaders ok")


if __name__ == "__main__":
    breakout()
    amidar()
    space_invaders()
    print("WORKER_OK")
