#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference tree (run in the build container only).

Inputs : /root/reference/toybox/interventions/defaults/<game>_{config,state}_default.json
         (true ctoybox dumps, written by code like toybox/interventions/space_invaders.py:187-197).
Outputs: tests/golden/rng_kat.json            -- RNG known-answer chain derived from those dumps
         tests/golden/<game>_config.json      -- the config dump, unchanged
         tests/golden/<game>_state.json       -- the state dump mapped to the ctoybox-0.5.0 key set
                                                 that toybox/interventions/*.py decode strictly
                                                 (interventions/base.py:209-225)

Only data is written: no reference source text is copied.  /root/reference does not exist on
the GPU box, so tests read the committed fixtures, never the reference tree.
"""
import json
import os
import sys

REF = os.environ.get("TOYBOX_REFERENCE", "/root/reference")
D = os.path.join(REF, "toybox", "interventions", "defaults")
OUT = os.path.dirname(os.path.abspath(__file__))
M = (1 << 64) - 1


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M


def rotr(x, k):
    return ((x >> k) | (x << (64 - k))) & M


def nxt(s):
    """xoroshiro128+ (55,14,36): returns (output, new_state)."""
    s0, s1 = s
    r = (s0 + s1) & M
    s1 ^= s0
    return r, (rotl(s0, 55) ^ s1 ^ ((s1 << 14) & M), rotl(s1, 36))


def prev(s):
    """inverse state transition"""
    s0n, s1n = s
    t = rotr(s1n, 36)
    s0 = rotr(s0n ^ t ^ ((t << 14) & M), 55)
    return (s0, t ^ s0)


def seed(n):
    return (0x193A6754A8A7D469 ^ n, 0x97830E05113BA7BB)


def child(p):
    a, p = nxt(p)
    b, p = nxt(p)
    return (a, b), p


def load(game, kind):
    with open(os.path.join(D, "%s_%s_default.json" % (game, kind))) as f:
        return json.load(f)


def rand_of(game, kind):
    return tuple(load(game, kind)["rand"]["state"])


def gen_range(s, n):
    """rand's UniformInt<u64>::sample_single widening-multiply rejection; returns (value, state, draws)."""
    zone = ((n << (64 - n.bit_length())) - 1) & M
    draws = []
    while True:
        v, s = nxt(s)
        draws.append(v)
        m = v * n
        if (m & M) <= zone:
            return m >> 64, s, draws


def make_rng_kat():
    kat = {}
    # KAT-A: Amidar -- config.rand == seed(13); state.rand == its first child
    p = seed(13)
    assert p == rand_of("amidar", "config")
    c, p_after = child(p)
    assert c == rand_of("amidar", "state")
    kat["amidar"] = {"seed": 13, "config_rand": list(p), "children_before_config": 0,
                     "state_rand": list(c), "state_draws_after_child": 0}
    # KAT-B: Breakout -- seed(13) -> child#1, child#2 -> parent == config.rand;
    #        child#2 advanced by the draws of one gen_range(4) == state.rand
    p = seed(13)
    c1, p = child(p)
    c2, p = child(p)
    assert p == rand_of("breakout", "config")
    idx, s_after, draws = gen_range(c2, 4)
    assert s_after == rand_of("breakout", "state"), "gen_range(4) must consume exactly the observed draws"
    assert len(draws) == 2 and idx == 2
    kat["breakout"] = {"seed": 13, "config_rand": list(p), "children_before_config": 2,
                       "child1": list(c1), "child2": list(c2),
                       "start_index": idx, "n_starts": 4, "range_draws": draws,
                       "state_rand": list(s_after)}
    # KAT-C: SpaceInvaders -- seed(17), two children; state.rand == child#2
    p = seed(17)
    c1, p = child(p)
    c2, p = child(p)
    assert p == rand_of("space_invaders", "config") and c2 == rand_of("space_invaders", "state")
    kat["space_invaders"] = {"seed": 17, "config_rand": list(p), "children_before_config": 2,
                             "child1": list(c1), "child2": list(c2), "state_rand": list(c2)}
    # raw generator outputs for a few seeds (self-consistency of ports)
    outs = {}
    for sd in (0, 13, 17, 1234, 2**31 - 1):
        s = seed(sd)
        o = []
        for _ in range(8):
            v, s = nxt(s)
            o.append(v)
        outs[str(sd)] = o
    kat["outputs"] = outs
    # gen_range known answers from seed(1234)
    s = seed(1234)
    gr = []
    for n in (2, 3, 4, 5, 6, 7, 10, 36, 100, 1000):
        v, s, d = gen_range(s, n)
        gr.append({"n": n, "value": v, "draws": len(d)})
    kat["gen_range_seed1234"] = gr
    return kat


def map_breakout_state(s):
    """old dump -> 0.5.0 key set (interventions/breakout.py:49-54): points->score, add level"""
    s = dict(s)
    s["score"] = s.pop("points")
    s["level"] = 1
    return s


def map_amidar_state(s):
    s = dict(s)
    s["level"] = 1
    return s


def map_si_state(s):
    """levels_completed->level; per-enemy move_* -> enemies_movement (interventions/space_invaders.py:16-19,116,146)"""
    s = json.loads(json.dumps(s))
    s["level"] = s.pop("levels_completed") + 1
    mc = s["enemies"][0]["move_counter"]
    right = s["enemies"][0]["move_right"]
    orient = s["enemies"][0]["orientation_init"]
    for e in s["enemies"]:
        for k in ("move_right", "move_down", "move_counter", "orientation_init"):
            e.pop(k)
    s["enemies_movement"] = {"move_counter": mc, "move_dir": "Right" if right else "Left",
                             "visual_orientation": bool(orient)}
    return s


def main():
    with open(os.path.join(OUT, "rng_kat.json"), "w") as f:
        json.dump(make_rng_kat(), f, indent=1)
    mappers = {"breakout": map_breakout_state, "amidar": map_amidar_state, "space_invaders": map_si_state}
    for game, mp in mappers.items():
        with open(os.path.join(OUT, "%s_config.json" % game), "w") as f:
            json.dump(load(game, "config"), f, separators=(",", ":"), sort_keys=True)
        with open(os.path.join(OUT, "%s_state.json" % game), "w") as f:
            json.dump(mp(load(game, "state")), f, separators=(",", ":"), sort_keys=True)
    # GridWorld: the two dumps as they are (the reference has no intervention classes for this game)
    for kind in ("config", "state"):
        with open(os.path.join(OUT, "gridworld_%s.json" % kind), "w") as f:
            json.dump(load("gridworld", kind), f, separators=(",", ":"), sort_keys=True)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    sys.exit(main())
