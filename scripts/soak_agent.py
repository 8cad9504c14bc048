#!/usr/bin/env python3
"""Agent-pipeline soak (GPU box): fused HIP observation path vs the CPU restatement with every wrapper on, thousands of
agent steps, so that resets of every kind, level changes and long frame-stack histories occur.  usage: soak_agent.py [steps] [envs]
SOAK_OBS=ring: the HIP engine keeps the plane ring (new_plane = 2) and its ring, read through the head index, is held to the
oracle's rolled stack."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import read_buffer, stack_from_ring, synthetic_actions  # noqa: E402
from toybox_amd import Engine, _abi  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
os.environ.setdefault("TBX_ORACLE_THREADS", "16")
olib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
_abi.bind(olib)
ring = os.environ.get("SOAK_OBS") == "ring"


def gpu_obs(g, returned):
    if not ring:
        return returned
    return stack_from_ring(read_buffer(g, _abi.BUF_AGENT_RING, (4, n, 84, 84)), g.agent_ring_head())


for game in ("breakout", "space_invaders", "amidar", "gridworld"):
    g, o = Engine(game, n), Engine(game, n, lib=olib)
    for e in (g, o):
        e.seed(777)
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=True, fire_reset=True, noop_max=30,
                     noop_seed=5, env_offset=1000, new_plane=2 if (ring and e is g) else 0)
    assert np.array_equal(gpu_obs(g, g.agent_reset()), o.agent_reset())
    t0 = time.time()
    dones = eps = 0
    for t in range(steps):
        a = synthetic_actions(game, n, t, seed=31)
        x, y = g.agent_step(a), o.agent_step(a)
        x = (gpu_obs(g, x[0]), x[1], x[2])
        for p, q, name in zip(x, y, ("obs", "reward", "done")):
            if not np.array_equal(p, q):
                print("%s: %s differs at agent step %d" % (game, name, t))
                sys.exit(1)
        eg, eo = g.agent_episodes(), o.agent_episodes()
        if not (np.array_equal(eg[0], eo[0]) and np.array_equal(eg[1][eg[0]], eo[1][eo[0]]) and np.array_equal(eg[2][eg[0]], eo[2][eo[0]])):
            print("%s: episode records differ at agent step %d" % (game, t))
            sys.exit(1)
        dones += int(x[2].sum())
        eps += int(eg[0].sum())
    for i in range(n):
        if bytes(g.get_state(i)) != bytes(o.get_state(i)):
            print("%s: state of env %d differs" % (game, i))
            sys.exit(1)
    print("%s%s: %d envs x %d agent steps identical (%d dones, %d finished games) in %.0f s" % (game, " (plane ring)" if ring else "", n, steps, dones, eps, time.time() - t0),
          flush=True)
    g.close()
    o.close()
print("agent soak ok")
