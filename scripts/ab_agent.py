#!/usr/bin/env python3
"""A/B timing of the agent step of several library builds on ONE box, interleaved.  usage: ab_agent.py game lib1.so lib2.so[:ring] ...
(":ring" behind a path: that arm runs with tbx_agent_config_t::new_plane = 2, the ring of planes instead of the rolled stack)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game = sys.argv[1]
libs = []
for spec in sys.argv[2:]:
    p = spec.split(":")[0]                                  # "lib.so[:ring][:formK]" (formK = TBX_OPT_STEP_FORM K for that arm)
    lib = C.CDLL(p)
    for name, (res, args) in _abi.PROTOTYPES.items():     # older builds lack the newest entry points: bind what is there
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    libs.append((spec, lib))
n = 65536
for rnd in range(3):
    for p, lib in libs:
        e = Engine(game, n, lib=lib)
        if ":form" in p:
            e.set_option(_abi.OPT_STEP_FORM, int(p.split(":form")[1].split(":")[0]))
        e.seed(1234)
        dm = bool(int(os.environ.get("AB_DEEPMIND", "0")))
        e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=dm, fire_reset=dm, noop_max=30 if dm else 0,
                     new_plane=2 if ":ring" in p else 0)
        e.agent_reset()
        for t in range(int(os.environ.get("AB_PREROLL", "10"))):
            e.agent_step_synthetic(1337, t)
        hip.synchronize()
        t0 = time.perf_counter()
        for t in range(1000, 1060):
            e.agent_step_synthetic(1337, t)
        hip.synchronize()
        dt = (time.perf_counter() - t0) / 60
        print("round %d %-30s %.4f ms  %.2f M agent-steps/s" % (rnd, p.split("/")[-1], dt * 1e3, n / dt / 1e6), flush=True)
        e.close()
