#!/usr/bin/env python3
"""Rasteriser launches of several library builds x several TBX_OPT_RENDER_SPLIT values on ONE box, interleaved: render-only and
[step ; render] loops.   python scripts/split_ab.py game envs "0,5,7,12" lib1.so lib2.so ...   (0 = the engine's choice)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game, n, splits = sys.argv[1], int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")]
libs = []
for p in sys.argv[4:]:
    lib = C.CDLL(p)
    for name, (res, args) in _abi.PROTOTYPES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    libs.append((os.path.basename(p), lib))
K = max(40, 40 * 65536 // n // 4)
engines = []
for name, lib in libs:
    e = Engine(game, n, lib=lib)
    e.seed(1234); e.new_game()
    for t in range(400):
        e.step_synthetic(1337, t)
    engines.append((name, e))
res = {}
for rnd in range(3):
    for name, e in engines:
        for sp in splits:
            e.set_option(_abi.OPT_RENDER_SPLIT, sp)
            for with_step in (0, 1):
                for k in range(5):
                    e.render_device(channels=3)
                hip.synchronize()
                t0 = time.perf_counter()
                for k in range(K):
                    if with_step:
                        e.step_synthetic(1337, 1000 + rnd * 100 + k)
                    e.render_device(channels=3)
                hip.synchronize()
                res.setdefault((name, sp, with_step), []).append(1000 * (time.perf_counter() - t0) / K)
fb = n * engines[0][1].height * engines[0][1].width * 3
for (name, sp, ws), v in sorted(res.items()):
    med = sorted(v)[len(v) // 2]
    print("%-24s split %2d %-13s %s  median %.4f ms  %.3f of 8 TB/s" % (name, sp, "[step;render]" if ws else "[render]", " ".join("%.4f" % x for x in v), med, fb / (med * 1e-3) / 8e12))
