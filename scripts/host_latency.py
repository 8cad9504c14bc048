import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from toybox_amd import Engine
for n in (1, 16, 256, 4096):
    e = Engine("breakout", n)
    a = np.ones(n, np.int32)
    for _ in range(50): e.step(a)
    t0 = time.perf_counter()
    for _ in range(500): e.step(a, auto_reset=True)
    dt = (time.perf_counter() - t0) / 500
    print("tbx_step host path n=%d: %.1f us" % (n, dt * 1e6))
    t0 = time.perf_counter()
    for _ in range(200): e.scalars()
    print("   scalars: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
    e.close()
