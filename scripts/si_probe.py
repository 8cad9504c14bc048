import sys, time
sys.path.insert(0, "/root/repo")
from toybox_amd import Engine, hip
n = 65536
def run(cfgmod, label):
    e0 = Engine("space_invaders", 1)
    cfg = e0.get_config(); e0.close()
    cfgmod(cfg)
    e = Engine("space_invaders", n, config=cfg)
    e.seed(1234)
    e.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True)
    e.agent_reset()
    for t in range(10): e.agent_step_synthetic(1337, t)
    hip.synchronize(); t0 = time.perf_counter()
    for t in range(10, 50): e.agent_step_synthetic(1337, t)
    hip.synchronize(); dt = (time.perf_counter() - t0) / 40
    print("%-20s %.3f ms" % (label, dt * 1e3), flush=True)
    e.close()
run(lambda c: None, "default")
def no_sh(c): c.n_shields = 0
run(no_sh, "no shields")
def one_row(c): c.n_rows = 1
run(one_row, "1 enemy row")
def both(c): c.n_rows = 1; c.n_shields = 0
run(both, "1 row, no shields")
run(lambda c: None, "default again")
