#!/usr/bin/env python3
"""Long soak of the overlapped rollout forms (GPU box): rollout chunks (tbx_rollout_synthetic, K = 4 record ring) and overlapped
fused launches (the device-side ticket) against the CPU restatement over tens of thousands of frames without a synchronisation in
between -- every chunk's step records compared through the gathered block, full states and frames at checkpoints, and tbx_sync's
report at the end (a ticket time-out of the wait kernel would show there).  usage: soak_rollout.py [frames] [envs] [game:form,...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import read_buffer  # noqa: E402
from toybox_amd import Engine, _abi, hip  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = 4
os.environ.setdefault("TBX_ORACLE_THREADS", "16")
olib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
_abi.bind(olib)
runs = [tuple(r.split(":")) for r in sys.argv[3].split(",")] if len(sys.argv) > 3 else \
    [("breakout", "chunks"), ("space_invaders", "chunks"), ("breakout", "ticket")]
for game, form in runs:
    g, o = Engine(game, n), Engine(game, n, lib=olib)
    for e in (g, o):
        e.seed(777)
        e.new_game()
        e.set_option(_abi.OPT_GATHER_EVERY, K)
        e.gather_init(1, 0, e.gather_unique_id())
    g.set_option(_abi.OPT_ROLLOUT_CHUNKS, int(os.environ.get("SR_FORM", _abi.ROLLOUT_CHUNKS_ON)))   # (3 / 4 / 5: the rasteriser forms by name)
    g.set_option(_abi.OPT_FUSED_OVERLAP, _abi.FUSED_OVERLAP_ON if form == "ticket" else _abi.FUSED_OVERLAP_OFF)
    st = hip.Stream()
    t0 = time.time()
    checks = 0
    for t in range(0, frames, K):
        if form == "chunks":
            g.rollout_synthetic(4242, t, K, channels=3, auto_reset=True, stream=st.ptr)
        else:
            for j in range(K):
                g.render_step_synthetic(4242, t + j, channels=3, auto_reset=True, stream=st.ptr)
                g.gather(stream=st.ptr)
        for j in range(K):                                                    # (the checker steps only: its frames are compared at the checkpoints)
            o.step_synthetic(4242, t + j, auto_reset=True)
            o.gather()
        if (t // K) % 50 == 49:                                               # every 200 frames: the last K steps' records of every env
            if not np.array_equal(g.gather_host().reshape(K, -1)[:, :n], o.gather_host().reshape(K, -1)[:, :n]):
                print("%s %s: step records differ in the chunk that ends at frame %d" % (game, form, t + K))
                sys.exit(1)
            checks += 1
        if (t // K) % 2500 == 2499:                                           # every 10 000 frames: states and the picture
            g.sync()
            for i in range(0, n, 41):
                if bytes(g.get_state(i)) != bytes(o.get_state(i)):
                    print("%s %s: state of env %d differs after %d frames" % (game, form, i, t + K))
                    sys.exit(1)
            if not np.array_equal(g.render(3)[:48], o.render(3)[:48]):
                print("%s %s: frames differ after %d frames" % (game, form, t + K))
                sys.exit(1)
    g.sync()                                                                  # raises on a ticket time-out
    for i in range(0, n, 17):
        assert bytes(g.get_state(i)) == bytes(o.get_state(i)), (game, form, i)
    sc, lv, le, ov = o.scalars()
    print("%s, %s: %d envs x %d frames identical (%d record checks, max level %d, max score %d) in %.0f s"
          % (game, form, n, frames, checks, int(le.max()), int(sc.max()), time.time() - t0), flush=True)
    g.close(); o.close()
print("soak ok")
