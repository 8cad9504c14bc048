#!/usr/bin/env python3
"""What a short kernel serialised between two rasteriser launches costs: one engine, 2-second blocks of render only /
step + render in stream order / step + render with TBX_OPT_PIPELINE = 2 (the step beside the previous render).
usage: interleave_probe.py [game]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

n = 65536
e = Engine(sys.argv[1] if len(sys.argv) > 1 else "breakout", n)
e.seed(1234)
e.new_game()
for t in range(600):
    e.step_synthetic(1337, t, auto_reset=True)
hip.synchronize()
T = [1000]
st = hip.Stream()


def render_only():
    e.render_device(channels=3, stream=st.ptr)


def step_render():
    e.step_synthetic(1337, T[0], auto_reset=True, stream=st.ptr)
    T[0] += 1
    e.render_device(channels=3, stream=st.ptr)


def block(name, fn, seconds=2.0):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(25):
            fn()
            k += 1
        hip.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.4f ms per pass" % (name, dt / k * 1e3), flush=True)


for rnd in range(3):
    block("render only", render_only)
    e.set_option(_abi.OPT_PIPELINE, 0)
    block("step + render, stream order", step_render)
    e.set_option(_abi.OPT_PIPELINE, 2)
    block("step + render, pipelined", step_render)
    e.set_option(_abi.OPT_PIPELINE, 0)
