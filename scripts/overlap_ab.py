#!/usr/bin/env python3
"""Same-box, same-process, interleaved A/B of TBX_OPT_FUSED_OVERLAP (round 6): for every batch size, Breakout engines with no
gather / a collective per step / a K-step ring, each pre-rolled, then rounds of [stream order, overlapped] x [no gather, K=1,
K=ring] of STEPS fused calls each -- ms per step of the bench's fused loop (tbx_render_step_synthetic [; tbx_gather]).
  python scripts/overlap_ab.py [sizes ...]    (env OA_ROUNDS, OA_STEPS, OA_RING, OA_CHANNELS, OA_LEADS = comma-separated
  TBX_OPT_FUSED_OVERLAP_LEAD values, 0 = the engine's choice)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [4096, 8192, 16384, 65536]
rounds, ring, C = int(os.environ.get("OA_ROUNDS", "5")), int(os.environ.get("OA_RING", "4")), int(os.environ.get("OA_CHANNELS", "3"))
leads = [int(v) for v in os.environ.get("OA_LEADS", "0").split(",")]
modes = [(_abi.FUSED_OVERLAP_OFF, 0, "order")] + [(_abi.FUSED_OVERLAP_ON, L, "overlap" if L == 0 else "ov%d" % L) for L in leads]
for n in sizes:
    K = int(os.environ.get("OA_STEPS", "0")) or max(200, min(2000, 200 * 65536 // n // 4))
    engines = {}
    for name, every in (("none", 0), ("k1", 1), ("k%d" % ring, ring)):
        e = Engine("breakout", n)
        e.seed(1234); e.new_game()
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

    def run(name, steps):
        e, g, t = engines[name], name != "none", ts[name]
        for _ in range(steps):
            e.render_step_synthetic(1337, t, channels=C, auto_reset=True, stream=st.ptr)
            if g:
                e.gather(stream=st.ptr)
            t += 1
        ts[name] = t

    for r in range(rounds):
        for name, e in engines.items():
            for mode, lead, label in modes:
                e.set_option(_abi.OPT_FUSED_OVERLAP, mode)
                e.set_option(_abi.OPT_FUSED_OVERLAP_LEAD, lead)
                run(name, 40)
                hip.synchronize()
                w0 = time.perf_counter()
                run(name, K)
                hip.synchronize()
                out.setdefault("%s_%s" % (label, name), []).append(1000.0 * (time.perf_counter() - w0) / K)
    line = {"envs": n, "steps": K, "channels": C}
    for k, v in out.items():
        med = sorted(v)[len(v) // 2]
        line[k] = {"ms": [round(x, 4) for x in v], "median": round(med, 4), "M_per_s": round(n / med / 1e3, 2),
                   "frac": round(n * engines[k.split("_", 1)[1]].height * engines[k.split("_", 1)[1]].width * C / med / 1e6 / 8000.0, 4)}
    for name in engines:
        for _, _, label in modes[1:]:
            line["gain_%s_%s" % (label, name)] = round(line["order_" + name]["median"] / line["%s_%s" % (label, name)]["median"] - 1.0, 4)
    line = {k: (v if not isinstance(v, dict) or os.environ.get("OA_VERBOSE") else {"median": v["median"], "frac": v["frac"]}) for k, v in line.items()}
    print(json.dumps(line), flush=True)
    for e in engines.values():
        e.sync()
        e.close()
