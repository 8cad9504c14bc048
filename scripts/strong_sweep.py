#!/usr/bin/env python3
"""Loop forms and record-ring depths against each other in ONE process on ONE box: for every batch size, engines with no
gather / a collective per step / a K-step ring, each pre-rolled, then rounds of [pair, fused] x [no gather, K=1, K=ring] of
STEPS steps each -- ms per step of the bench loop (one frame stepped + one RGB frame rasterised per env).
  python scripts/strong_sweep.py [game] [sizes ...]    (env SS_ROUNDS, SS_STEPS, SS_RING, SS_LIB, SS_STEP_FORM)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "breakout"
sizes = [int(v) for v in sys.argv[2:]] or [8192, 65536]
rounds, ring = int(os.environ.get("SS_ROUNDS", "4")), int(os.environ.get("SS_RING", "4"))
LIB = None
if os.environ.get("SS_LIB"):                       # another build of the library (A/B in one call: run the script twice on one box)
    import ctypes
    LIB = _abi.bind(ctypes.CDLL(os.path.abspath(os.environ["SS_LIB"])), older_build=True)
for n in sizes:
    K = int(os.environ.get("SS_STEPS", "0")) or max(200, min(2000, 200 * 65536 // n // 4))
    engines = {}
    for name, every in (("none", 0), ("k1", 1), ("k%d" % ring, ring)):
        e = Engine(game, n, lib=LIB)
        e.seed(1234); e.new_game()
        if os.environ.get("SS_STEP_FORM"):             # Amidar: 2 = the wave-per-env step (and with it the fused launch) at every size
            e.set_option(_abi.OPT_STEP_FORM, int(os.environ["SS_STEP_FORM"]))
        if every:
            e.set_option(_abi.OPT_GATHER_EVERY, every)
            e.gather_init(1, 0, e.gather_unique_id())
        engines[name] = e
    st = hip.Stream()
    ts = {}
    for name, e in engines.items():
        for t in range(600):
            e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
        ts[name] = 600
    out = {}

    def run(name, fused, steps):
        e, g, t = engines[name], name != "none", ts[name]
        for _ in range(steps):
            if fused:
                e.render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=st.ptr)
                if g:
                    e.gather(stream=st.ptr)
            else:
                e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
                if g:
                    e.gather(stream=st.ptr)
                e.render_device(0, 3, stream=st.ptr)
            t += 1
        ts[name] = t

    for r in range(rounds):
        for name in engines:
            for fused in (False, True):
                run(name, fused, 30)
                hip.synchronize()
                w0 = time.perf_counter()
                run(name, fused, K)
                hip.synchronize()
                out.setdefault("%s_%s" % ("fused" if fused else "pair", name), []).append(1000.0 * (time.perf_counter() - w0) / K)
    line = {"game": game, "envs": n, "steps": K}
    for k, v in out.items():
        med = sorted(v)[len(v) // 2]
        line[k] = {"ms": [round(x, 4) for x in v], "median": round(med, 4), "M_per_s": round(n / med / 1e3, 2)}
    print(json.dumps(line), flush=True)
    for e in engines.values():
        e.close()
