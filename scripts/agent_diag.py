#!/usr/bin/env python3
"""What the fused agent observation kernel's time is made of: the DIAG build of the library (make DIAG=1 ->
scripts/ab/lib_diag.so) runs the agent step with parts of agent_fused_wave switched off by TBX_AGENT_DIAG (observations are
WRONG with any bit set: a measurement build).  One process per variant (the knob is read once per launch from the environment);
rounds interleaved.  usage: agent_diag.py [game] [envs]          -> ms per agent step per variant
With AGENT_DIAG_ONE=<bits> it runs that one variant for rocprofv3 --pmc (scripts/agent_diag.sh); AGENT_DIAG_OBS=ring: the plane
ring (new_plane = 2) instead of the rolled stack."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
game = sys.argv[1] if len(sys.argv) > 1 else "space_invaders"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
VARIANTS = [(0, "full kernel"), (1, "no stack commit"), (2, "set-up only (no scanline loop)"), (4, "scanline loop, every row skipped"),
            (8, "one painter (never frame A)"), (16, "no fast rows (enemy-only scanlines painted)"), (3, "set-up only, no commit")]


def one(bits, steps=40):
    from toybox_amd import Engine, _abi, hip
    lib = _abi.bind(ctypes.CDLL(os.path.join(ROOT, "scripts", "ab", "lib_diag.so")))
    e = Engine(game, n, lib=lib)
    e.seed(1234)
    e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, new_plane=2 if os.environ.get("AGENT_DIAG_OBS") == "ring" else 0)
    e.agent_reset()
    st = hip.Stream()
    for t in range(60):                                   # mid-game states (observations do not feed back into the games)
        e.agent_step_synthetic(1337, t, stream=st.ptr)
    hip.synchronize()
    t0 = time.perf_counter()
    for t in range(60, 60 + steps):
        e.agent_step_synthetic(1337, t, stream=st.ptr)
    hip.synchronize()
    dt = (time.perf_counter() - t0) / steps
    e.close()
    return dt * 1e3


if os.environ.get("AGENT_DIAG_ONE") is not None:
    os.environ["TBX_AGENT_DIAG"] = os.environ["AGENT_DIAG_ONE"]
    print(json.dumps({"bits": int(os.environ["AGENT_DIAG_ONE"]), "ms": one(int(os.environ["AGENT_DIAG_ONE"]), steps=12)}))
    sys.exit(0)

res = {b: [] for b, _ in VARIANTS}
for rnd in range(3):
    for b, _ in VARIANTS:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), game, str(n)], env=dict(os.environ, AGENT_DIAG_ONE=str(b)),
                           capture_output=True, text=True, timeout=600)
        try:
            res[b].append(json.loads(p.stdout.strip().splitlines()[-1])["ms"])
        except Exception:
            print("variant %d failed: %s" % (b, (p.stdout + p.stderr)[-500:]), file=sys.stderr)
print("%s, %d envs%s: ms per agent step (4 frames + observation), three rounds" % (game, n, ", plane ring" if os.environ.get("AGENT_DIAG_OBS") == "ring" else ""))
for b, name in VARIANTS:
    print("  diag %2d  %-46s %s" % (b, name, "  ".join("%.3f" % v for v in res[b])))
