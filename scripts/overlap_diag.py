#!/usr/bin/env python3
"""What an overlapped fused launch pays for, part by part: the DIAG build (scripts/build_diag.sh -> scripts/ab/lib_diag.so) with
TBX_OVERLAP_DIAG masks against stream order, interleaved in one process.  Results with a mask are not valid frames / states --
this measures time only.   python scripts/overlap_diag.py [sizes ...]   (env OD_ROUNDS, OD_MASKS, OD_LEAD)"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toybox_amd import Engine, _abi, hip  # noqa: E402

lib = None if os.environ.get("OD_LIB") == "product" else _abi.bind(ctypes.CDLL(os.path.join(ROOT, "scripts", "ab", "lib_diag.so")))
sizes = [int(v) for v in sys.argv[1:]] or [8192, 65536]
rounds = int(os.environ.get("OD_ROUNDS", "3"))
masks = [int(v) for v in os.environ.get("OD_MASKS", "0,8,10,9,12,24,40,72,136,255,2,1,16,32,3,19").split(",")]
for n in sizes:
    K = max(150, min(1500, 150 * 65536 // n // 4))
    e = Engine("breakout", n, lib=lib)
    e.seed(1234); e.new_game()
    e.set_option(_abi.OPT_FUSED_OVERLAP_LEAD, int(os.environ.get("OD_LEAD", "0")))
    G = int(os.environ.get("OD_GATHER", "0"))
    if G:
        e.set_option(_abi.OPT_GATHER_EVERY, G)
        e.gather_init(1, 0, e.gather_unique_id())
    st = hip.Stream()
    t = 0
    for _ in range(600):
        e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr); t += 1
    out = {}
    for r in range(rounds):
        for label, mode, mask in [("order", _abi.FUSED_OVERLAP_OFF, 0)] + [("m%d" % m, _abi.FUSED_OVERLAP_ON, m) for m in masks]:
            os.environ["TBX_OVERLAP_DIAG"] = str(mask)
            e.set_option(_abi.OPT_FUSED_OVERLAP, mode)
            for phase in range(2):
                hip.synchronize()
                w0 = time.perf_counter()
                for _ in range(30 if phase == 0 else K):
                    e.render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=st.ptr); t += 1
                    if G:
                        e.gather(stream=st.ptr)
                hip.synchronize()
            out.setdefault(label, []).append(1000.0 * (time.perf_counter() - w0) / K)
    base = sorted(out["order"])[len(out["order"]) // 2]
    print(json.dumps({"lib": os.environ.get("OD_LIB", "diag"), "envs": n, "steps": K, "gather": G, "lane_priority": os.environ.get("TBX_LANE_PRIORITY", "high"), **{k: [round(sorted(v)[len(v) // 2], 4), round(sorted(v)[len(v) // 2] / base - 1.0, 4)] for k, v in out.items()}}), flush=True)
    try:
        e.sync()
    except Exception as ex:
        print("sync:", ex)
    e.close()
