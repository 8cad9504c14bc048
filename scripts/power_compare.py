#!/usr/bin/env python3
"""Socket power (rocm-smi) under three loops of ~6 s each on one box: hipMemsetAsync of the Breakout frame batch, the Breakout
rasteriser alone, the Breakout step + rasteriser -- with the rate each reaches.  What the board draws for the same bytes."""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, hip  # noqa: E402

n = 65536
e = Engine("breakout", n)
e.seed(1234)
e.new_game()
for t in range(600):
    e.step_synthetic(1337, t, auto_reset=True)
nbytes = n * e.height * e.width * 3
rt = hip.runtime()
rt.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
buf = hip.malloc(nbytes)


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            w = re.search(r"Power \(W\): ([0-9.]+)", txt)
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            mem = re.search(r"Sensor memory\) \(C\): ([0-9.]+)", txt)
            if w:
                out.append((float(w.group(1)), int(sclk.group(1)) if sclk else 0, float(mem.group(1)) if mem else 0.0))
        except Exception:
            pass
        time.sleep(0.3)


def run(name, fn, seconds=6.0):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    hip.synchronize()
    th.start()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn(k)
            k += 1
        hip.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    ws = [o[0] for o in out[2:]] or [0.0]
    print("%-28s %.3f ms per pass  %5.0f GB/s   power W min/median/max %4.0f / %4.0f / %4.0f   sclk %s MHz   memory %s C" % (
        name, dt / k * 1e3, nbytes * k / dt / 1e9, min(ws), sorted(ws)[len(ws) // 2], max(ws),
        sorted(o[1] for o in out)[len(out) // 2] if out else "?", max(o[2] for o in out) if out else "?"), flush=True)
    time.sleep(2.0)


run("hipMemsetAsync 7.55 GB", lambda k: hip.check(rt.hipMemsetAsync(buf, k & 255, nbytes, None), "memset"))
run("breakout render (RGB)", lambda k: e.render_device(channels=3))
run("breakout step + render", lambda k: (e.step_synthetic(1337, 1000 + k, auto_reset=True), e.render_device(channels=3)))
run("hipMemsetAsync 7.55 GB again", lambda k: hip.check(rt.hipMemsetAsync(buf, k & 255, nbytes, None), "memset"))
