#!/usr/bin/env python3
"""The random-rollout loop in its three forms on ONE engine, interleaved on one box: single fused calls in stream order
(`order`), single fused calls overlapped behind the device-side ticket (`ticket`, TBX_OPT_FUSED_OVERLAP), and rollout chunks of
k steps (`chunks`; tbx_rollout_synthetic / TBX_OPT_ROLLOUT_CHUNKS with the rasteriser form the engine's choice; `frames` / `span`: the
form named -- a rasteriser launch per frame on two lanes / one per chunk on one; `pair` / `pipe` = the two-launch
loop in stream order / with TBX_OPT_PIPELINE = the engine's choice) -- ms per step, with the K = k record ring (1-rank
communicator) or without a gather.   python scripts/rollout_ab.py [sizes ...]   (env RA_ROUNDS, RA_K, RA_GATHER = 0 / 1, RA_FORMS)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toybox_amd import Engine, _abi, hip  # noqa: E402

LIB = None
if os.environ.get("RA_LIB"):                       # another build (the DIAG build: TBX_LANE_PRIORITY)
    import ctypes
    LIB = _abi.bind(ctypes.CDLL(os.path.abspath(os.environ["RA_LIB"])), older_build=True)

sizes = [int(v) for v in sys.argv[1:]] or [4096, 8192]
rounds, k, G = int(os.environ.get("RA_ROUNDS", "5")), int(os.environ.get("RA_K", "4")), int(os.environ.get("RA_GATHER", "1"))
forms = os.environ.get("RA_FORMS", "order,ticket,frames,span").split(",")
GAME = os.environ.get("RA_GAME", "breakout")
for n in sizes:
    K = max(40, min(400, 40 * 65536 // n // 4)) * k              # steps per timed region, a multiple of k
    e = Engine(GAME, n, lib=LIB)
    e.seed(1234); e.new_game()
    if G:
        e.set_option(_abi.OPT_GATHER_EVERY, k)
        e.gather_init(1, 0, e.gather_unique_id())
    st = hip.Stream()
    t = 0
    for _ in range(600):
        e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr); t += 1
    out = {}

    def run(form, steps):
        global t
        if form in ("pair", "pipe"):                             # the policy loop: two launches per step (pipe: TBX_OPT_PIPELINE = the engine's choice)
            for _ in range(steps):
                e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr); t += 1
                if G:
                    e.gather(stream=st.ptr)
                e.render_device(0, 3, stream=st.ptr)
        elif form in ("chunks", "frames", "span"):
            for _ in range(steps // k):
                e.rollout_synthetic(1337, t, k, channels=3, auto_reset=True, stream=st.ptr); t += k
        else:
            for _ in range(steps):
                e.render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=st.ptr); t += 1
                if G:
                    e.gather(stream=st.ptr)

    for r in range(rounds):
        for form in forms:
            e.set_option(_abi.OPT_FUSED_OVERLAP, _abi.FUSED_OVERLAP_ON if form == "ticket" else _abi.FUSED_OVERLAP_OFF)
            e.set_option(_abi.OPT_ROLLOUT_CHUNKS, {"frames": _abi.ROLLOUT_CHUNKS_PER_FRAME, "span": _abi.ROLLOUT_CHUNKS_SPAN}.get(form, _abi.ROLLOUT_CHUNKS_ON))
            e.set_option(_abi.OPT_PIPELINE, 1 if form == "pipe" else 0)
            run(form, 10 * k)
            hip.synchronize()
            w0 = time.perf_counter()
            run(form, K)
            hip.synchronize()
            out.setdefault(form, []).append(1000.0 * (time.perf_counter() - w0) / K)
    H, W = e.height, e.width
    base = sorted(out[forms[0]])[len(out[forms[0]]) // 2]
    line = {"game": GAME, "envs": n, "k": k, "gather_ring": bool(G), "steps": K, "lib": os.environ.get("RA_LIB", "product"), "lane_priority": os.environ.get("TBX_LANE_PRIORITY", "high")}
    for f, v in out.items():
        med = sorted(v)[len(v) // 2]
        line[f] = {"median": round(med, 4), "min": round(min(v), 4), "max": round(max(v), 4), "vs_first": round(med / base - 1.0, 4),
                   "frac": round(n * H * W * 3 / med / 1e6 / 8000.0, 4)}
    print(json.dumps(line), flush=True)
    e.sync(); e.close()
