#!/usr/bin/env python3
"""Two builds of the library against each other on the bench's fused loop (tbx_render_step_synthetic [; tbx_gather]), interleaved in
one process on one box: ms per step, median of the rounds.
   python scripts/fused_lib_ab.py libA.so libB.so [sizes ...]      (env FA_ROUNDS, FA_GATHER = K of a 1-rank gather, FA_OVERLAP = option value for libs that know it)"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toybox_amd import Engine, _abi, hip  # noqa: E402

paths = sys.argv[1:3]
sizes = [int(v) for v in sys.argv[3:]] or [65536]
rounds, G = int(os.environ.get("FA_ROUNDS", "5")), int(os.environ.get("FA_GATHER", "0"))
libs = [_abi.bind(ctypes.CDLL(os.path.abspath(p)), older_build=True) for p in paths]
for n in sizes:
    K = max(100, min(1500, 150 * 65536 // n // 4))
    engines = []
    st = hip.Stream()
    for lib in libs:
        e = Engine("breakout", n, lib=lib)
        e.seed(1234); e.new_game()
        if "FA_OVERLAP" in os.environ:
            try:
                e.set_option(_abi.OPT_FUSED_OVERLAP, int(os.environ["FA_OVERLAP"]))
            except Exception:
                pass
        if G:
            e.set_option(_abi.OPT_GATHER_EVERY, G)
            e.gather_init(1, 0, e.gather_unique_id())
        for t in range(600):
            e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
        engines.append(e)
    out = [[] for _ in libs]
    t = 600
    for r in range(rounds):
        for k, e in enumerate(engines):
            for phase in range(2):
                hip.synchronize()
                w0 = time.perf_counter()
                for _ in range(30 if phase == 0 else K):
                    e.render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=st.ptr); t += 1
                    if G:
                        e.gather(stream=st.ptr)
                hip.synchronize()
            out[k].append(1000.0 * (time.perf_counter() - w0) / K)
    med = [sorted(v)[len(v) // 2] for v in out]
    print(json.dumps({"envs": n, "gather": G, "steps": K, "libs": paths, "ms": [[round(x, 4) for x in v] for v in out],
                      "median": [round(m, 4) for m in med], "B_over_A": round(med[1] / med[0] - 1.0, 4)}), flush=True)
    for e in engines:
        e.sync(); e.close()
