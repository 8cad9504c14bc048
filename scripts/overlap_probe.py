#!/usr/bin/env python3
"""Upper bound of what overlapping consecutive fused rollout launches could buy at small batches: TWO independent engines of n
envs each, their fused calls issued alternately on two streams, against ONE engine's loop on one stream (same total frames).
usage: overlap_probe.py [game] [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, hip  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "breakout"
for n in [int(v) for v in sys.argv[2:]] or [4096, 8192]:
    es = [Engine(game, n) for _ in range(2)]
    ss = [hip.Stream() for _ in range(2)]
    for k, e in enumerate(es):
        e.seed(1234 + k); e.new_game()
        for t in range(600):
            e.step_synthetic(1337, t, auto_reset=True, stream=ss[k].ptr)
    hip.synchronize()
    K = max(200, 200 * 65536 // n // 8)

    def one(t0):
        for t in range(t0, t0 + 2 * K):
            es[0].render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=ss[0].ptr)

    def two(t0):
        for t in range(t0, t0 + K):
            es[0].render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=ss[0].ptr)
            es[1].render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=ss[1].ptr)

    res = {"one": [], "two": []}
    t = 1000
    for rnd in range(4):
        for name, fn in (("one", one), ("two", two)):
            fn(t); t += 2 * K
            hip.synchronize()
            w = time.perf_counter()
            fn(t); t += 2 * K
            hip.synchronize()
            res[name].append(1e3 * (time.perf_counter() - w) / (2 * K))
    print(game, n, {k: round(sorted(v)[len(v) // 2], 4) for k, v in res.items()}, "ms per fused call (one stream / two engines alternating on two streams)", flush=True)
    for e in es:
        e.close()
