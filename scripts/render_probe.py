#!/usr/bin/env python3
"""Times the render launch of one game on one box: render_probe.py game channels [envs] [preroll] [split ...] -- one line per
waves-per-frame value (TBX_OPT_RENDER_SPLIT; none given: the engine's choice).  TBX_SI_DIAG / TBX_SI_NO_SKIP only act on a
measurement build of the library (make -C toybox_amd/csrc DIAG=1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game, ch = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 400
e = Engine(game, n)
e.seed(1234)
e.new_game()
for t in range(pre):
    e.step_synthetic(1337, t)
for split in ([int(v) for v in sys.argv[5:]] or [0]):
    e.set_option(_abi.OPT_RENDER_SPLIT, split)
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
    print("%-14s ch=%d n=%-6d split=%-2d %-30s %.4f ms  %.0f GB/s" % (game, ch, n, split, tag, best * 1e3, n * e.height * e.width * ch / best / 1e9), flush=True)
