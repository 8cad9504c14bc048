#!/usr/bin/env python3
"""A/B timing of render launches of several builds of the library on ONE box, interleaved (run-to-run drift on a box is a
few per cent, so variants are only comparable within one session).  usage: ab_render.py game channels lib1.so lib2.so ..."""
import ctypes as C
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game, ch = sys.argv[1], int(sys.argv[2])
libs = []
for spec in sys.argv[3:]:                                  # "lib.so:formK" = TBX_OPT_STEP_FORM K for that arm
    p = spec.split(":form")[0]
    lib = C.CDLL(p)
    for name, (res, args) in _abi.PROTOTYPES.items():     # older builds lack the newest entry points: bind what is there
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    libs.append((spec, lib))
n = int(__import__("os").environ.get("AB_ENVS", "65536"))
pre = int(__import__("os").environ.get("AB_PREROLL", "30"))     # frames played before timing (mid-game states paint more)
with_step = __import__("os").environ.get("AB_STEP", "0") == "1"   # time [step ; render] instead of [render]
for rnd in range(3):
    for p, lib in libs:
        e = Engine(game, n, lib=lib)
        if __import__("os").environ.get("AB_SPLIT"):              # waves per frame (TBX_OPT_RENDER_SPLIT); "a,b": one value per library
            sp = __import__("os").environ["AB_SPLIT"].split(",")
            e.set_option(_abi.OPT_RENDER_SPLIT, int(sp[min(len(sp) - 1, [q for q, _ in libs].index(p))]))
        if ":form" in p:
            e.set_option(_abi.OPT_STEP_FORM, int(p.split(":form")[1]))
        e.seed(1234)
        for t in range(pre):
            e.step_synthetic(1337, t)
        e.render_device(channels=ch)
        hip.synchronize()
        t0 = time.perf_counter()
        for k in range(40):
            if with_step:
                e.step_synthetic(1337, pre + k)
            e.render_device(channels=ch)
        hip.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print("round %d %-40s %.4f ms  %.0f GB/s" % (rnd, p.split("/")[-1], dt * 1e3, n * e.height * e.width * ch / dt / 1e9), flush=True)
        e.close()
