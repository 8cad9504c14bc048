#!/usr/bin/env python3
"""Cost probe of the SpaceInvaders fused agent kernel: what do shields, enemy rows and the second painter cost? (GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toybox_amd import Engine, hip  # noqa: E402

n = 65536


def run(cfgmod, label, skip=4):
    e0 = Engine("space_invaders", 1)
    cfg = e0.get_config()
    e0.close()
    cfgmod(cfg)
    e = Engine("space_invaders", n, config=cfg)
    e.seed(1234)
    e.agent_init(skip=skip, out_h=84, out_w=84, stack=4, clip_reward=True)
    e.agent_reset()
    for t in range(10):
        e.agent_step_synthetic(1337, t)
    hip.synchronize()
    t0 = time.perf_counter()
    for t in range(10, 50):
        e.agent_step_synthetic(1337, t)
    hip.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("%-28s %.3f ms" % (label, dt * 1e3), flush=True)
    e.close()


def no_sh(c):
    c.n_shields = 0


def one_row(c):
    c.n_rows = 1


def both(c):
    c.n_rows = 1
    c.n_shields = 0


run(lambda c: None, "default skip 4")
run(no_sh, "no shields")
run(one_row, "1 enemy row")
run(both, "1 row, no shields")
run(lambda c: None, "default skip 2", skip=2)
run(lambda c: None, "default skip 1 (B only)", skip=1)
run(both, "1 row no shields, skip 1", skip=1)
