#!/usr/bin/env python3
"""One engine, one loop form, N iterations -- the thing to put under rocprofv3 --kernel-trace when the question is what happens
BETWEEN the kernels of the loop.   python scripts/loop_once.py game envs pair|fused|chunks|render|step iterations
env LO_OVERLAP = TBX_OPT_FUSED_OVERLAP (0 engine's choice, 1 overlapped, 2 stream order), LO_GATHER = K of a 1-rank record gather (0: none),
LO_CHUNKS = TBX_OPT_ROLLOUT_CHUNKS (0 engine's choice, 1 on, 2 off)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, _abi, hip  # noqa: E402

game, n, form, iters = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
e = Engine(game, n)
e.seed(1234); e.new_game()
e.set_option(_abi.OPT_FUSED_OVERLAP, int(os.environ.get("LO_OVERLAP", "0")))
e.set_option(_abi.OPT_ROLLOUT_CHUNKS, int(os.environ.get("LO_CHUNKS", "0")))
G = int(os.environ.get("LO_GATHER", "0"))
if G:
    e.set_option(_abi.OPT_GATHER_EVERY, G)
    e.gather_init(1, 0, e.gather_unique_id())
st = hip.Stream()
for t in range(300):
    e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
hip.synchronize()
for t in range(300, 300 + iters):
    if form == "pair":
        e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
        e.render_device(0, 3, stream=st.ptr)
    elif form == "fused":
        e.render_step_synthetic(1337, t, channels=3, auto_reset=True, stream=st.ptr)
        if G:
            e.gather(stream=st.ptr)
    elif form == "chunks":                     # tbx_rollout_synthetic, 4 frames per call (with LO_GATHER: the K = 4 ring)
        if (t - 300) % 4 == 0:
            e.rollout_synthetic(1337, t, 4, channels=3, auto_reset=True, stream=st.ptr)
    elif form == "render":
        e.render_device(0, 3, stream=st.ptr)
    else:
        e.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
hip.synchronize()
e.close()
