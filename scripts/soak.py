#!/usr/bin/env python3
"""Long-horizon parity soak (GPU box): HIP engine vs the CPU restatement over many thousands of frames with auto-reset, so
that late-game situations (fast formations, many levels, chase chains) are reached.  usage: soak.py [frames] [envs]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import synthetic_actions  # noqa: E402
from toybox_amd import Engine, _abi  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
os.environ.setdefault("TBX_ORACLE_THREADS", "16")
olib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
_abi.bind(olib)
for game in ("breakout", "space_invaders", "amidar", "gridworld"):
    g, o = Engine(game, n), Engine(game, n, lib=olib)
    for e in (g, o):
        e.seed(4242)
        e.new_game()
    t0 = time.time()
    dones = 0
    for t in range(frames):
        a = synthetic_actions(game, n, t, seed=99)
        rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)
        for x, y, name in zip(rg, ro, ("reward", "done", "lives", "score")):
            if not np.array_equal(x, y):
                print("%s: %s differs at frame %d, envs %s" % (game, name, t, np.nonzero(x != y)[0][:8]))
                sys.exit(1)
        dones += int(rg[1].sum())
    for i in range(0, n, 37):
        if bytes(g.get_state(i)) != bytes(o.get_state(i)):
            print("%s: state of env %d differs after %d frames" % (game, i, frames))
            sys.exit(1)
    if not np.array_equal(g.render(3)[:64], o.render(3)[:64]):
        print("%s: frames differ" % game)
        sys.exit(1)
    sc, lv, le, ov = o.scalars()
    print("%s: %d envs x %d frames identical (%d episode ends, max level %d, max score %d) in %.0f s"
          % (game, n, frames, dones, int(le.max()), int(sc.max()), time.time() - t0), flush=True)
    g.close()
    o.close()
print("soak ok")
