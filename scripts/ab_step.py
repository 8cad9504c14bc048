#!/usr/bin/env python3
"""A/B timing of the batch STEP (no render) of several library builds on ONE box, interleaved.
usage: ab_step.py game lib1.so lib2.so ...   (AB_ENVS, AB_PREROLL, AB_STEPS)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game = sys.argv[1]
libs = []
for spec in sys.argv[2:]:                                  # "lib.so:formK" = TBX_OPT_STEP_FORM K for that arm
    p = spec.split(":form")[0]
    lib = C.CDLL(p)
    for name, (res, args) in _abi.PROTOTYPES.items():     # older builds lack the newest entry points: bind what is there
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    libs.append((spec, lib))
n = int(os.environ.get("AB_ENVS", "65536"))
steps = int(os.environ.get("AB_STEPS", "400"))
for rnd in range(3):
    for p, lib in libs:
        e = Engine(game, n, lib=lib)
        if ":form" in p:
            e.set_option(_abi.OPT_STEP_FORM, int(p.split(":form")[1]))
        e.seed(1234)
        e.new_game()
        for t in range(int(os.environ.get("AB_PREROLL", "300"))):
            e.step_synthetic(1337, t, auto_reset=True)
        hip.synchronize()
        t0 = time.perf_counter()
        for t in range(1000, 1000 + steps):
            e.step_synthetic(1337, t, auto_reset=True)
        hip.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("round %d %-30s %8.2f us/step  %.1f M env-steps/s (step only)" % (rnd, p.split("/")[-1], dt * 1e6, n / dt / 1e6), flush=True)
        e.close()
