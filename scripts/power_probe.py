#!/usr/bin/env python3
"""Time series of the Breakout step + RGB render over ~10 s in chunks of 50 steps: shows the GPU's two rate states
(DESIGN.md section 6).  usage: power_probe.py [sleep_ms_between_chunks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, hip  # noqa: E402

pause = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.0
n = 65536
e = Engine("breakout", n)
e.seed(1234)
e.new_game()
for t in range(1000):
    e.step_synthetic(1337, t, auto_reset=True)
hip.synchronize()
t = 1000
t_start = time.perf_counter()
line = []
for chunk in range(160):
    t0 = time.perf_counter()
    for _ in range(50):
        e.step_synthetic(1337, t, auto_reset=True)
        e.render_device(channels=3)
        t += 1
    hip.synchronize()
    dt = (time.perf_counter() - t0) / 50
    line.append("%.3f" % (dt * 1e3))
    if len(line) == 20:
        print("t=%5.1fs  ms/step: %s" % (time.perf_counter() - t_start, " ".join(line)), flush=True)
        line = []
    if pause:
        time.sleep(pause)
