#!/usr/bin/env python3
"""Times the render launch of one game on one box: render_probe.py game channels [envs] [preroll].  Diagnostic env switches
(TBX_RENDER_SPLIT, TBX_SI_DIAG, ...) are read by the library at first use, so each setting is its own process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, hip  # noqa: E402

game, ch = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 400
e = Engine(game, n)
e.seed(1234)
e.new_game()
for t in range(pre):
    e.step_synthetic(1337, t)
for _ in range(5):
    e.render_device(channels=ch)
hip.synchronize()
best = 1e9
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(40):
        e.render_device(channels=ch)
    hip.synchronize()
    best = min(best, (time.perf_counter() - t0) / 40)
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("TBX_"))
print("%-14s ch=%d n=%-6d %-40s %.4f ms  %.0f GB/s" % (game, ch, n, tag, best * 1e3, n * e.height * e.width * ch / best / 1e9), flush=True)
