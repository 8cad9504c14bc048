#!/usr/bin/env python3
"""TBX_OPT_PIPELINE modes against each other in ONE process on ONE box (boxes differ by 5-15 %): for every batch size one
engine, pre-rolled, then rounds of [mode 0, mode 2, mode 3] x K steps each -- ms per step of the bench loop (step + RGB render
into the engine-owned frame buffer), optionally with the one-rank record gather queued between the two.
  python scripts/pipeline_sweep.py [game] [sizes ...]   (env PS_GATHER=1, PS_ROUNDS, PS_STEPS)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "breakout"
sizes = [int(v) for v in sys.argv[2:]] or [4096, 8192, 16384, 65536]
rounds, gather = int(os.environ.get("PS_ROUNDS", "4")), bool(os.environ.get("PS_GATHER"))
modes = [int(v) for v in os.environ.get("PS_MODES", "0,2,3").split(",")]
lib = None
if os.environ.get("PS_LIB"):                       # another build of the library (A/B in one call)
    import ctypes
    lib = _abi.bind(ctypes.CDLL(os.path.abspath(os.environ["PS_LIB"])), older_build=True)
res = {}
for n in sizes:
    K = int(os.environ.get("PS_STEPS", "0")) or max(200, min(2000, 200 * 65536 // n // 4))
    e = Engine(game, n, lib=lib)
    e.seed(1234); e.new_game()
    if os.environ.get("PS_STEP_FORM"):
        e.set_option(_abi.OPT_STEP_FORM, int(os.environ["PS_STEP_FORM"]))
    if gather:
        e.gather_init(1, 0, e.gather_unique_id())
    st = hip.Stream()
    t = 0
    for _ in range(600):
        e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr); t += 1
    fb = n * e.height * e.width * 3
    out = {m: [] for m in modes}
    for r in range(rounds):
        for m in modes:
            e.set_option(_abi.OPT_PIPELINE, m)
            for _ in range(30):
                e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
                if gather:
                    e.gather(stream=st.ptr)
                e.render_device(0, 3, stream=st.ptr); t += 1
            hip.synchronize()
            w0 = time.perf_counter()
            for _ in range(K):
                e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
                if gather:
                    e.gather(stream=st.ptr)
                e.render_device(0, 3, stream=st.ptr); t += 1
            hip.synchronize()
            out[m].append(1000.0 * (time.perf_counter() - w0) / K)
    e.sync()
    line = {"game": game, "envs": n, "steps": K, "gather": gather, "lib": os.environ.get("PS_LIB", "default"), "step_form": os.environ.get("PS_STEP_FORM", "auto")}
    for m in modes:
        best, med = min(out[m]), sorted(out[m])[len(out[m]) // 2]
        line["mode%d" % m] = {"ms_per_step": [round(v, 4) for v in out[m]], "median_ms": round(med, 4),
                              "Msteps_per_s": round(n / med / 1e3, 2), "frac_of_8TBs": round(fb / (med * 1e-3) / 8e12, 3)}
    print(json.dumps(line), flush=True)
    e.close()
