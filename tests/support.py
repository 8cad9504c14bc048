"""Helpers shared by the test-suite (host-side numpy restatements of tiny pure functions)."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

LEGAL = {
    "breakout": [0, 1, 3, 4],
    "amidar": [0, 1, 2, 3, 4, 5],
    "space_invaders": [0, 1, 3, 4, 11, 12],
    "gridworld": [0, 2, 3, 4, 5],
}


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def synthetic_actions(game, n, t, seed=1337, env_offset=0):
    """a = legal[ splitmix64(seed ^ (env << 32) ^ t) mod n_legal ]  (same rule as tbx_step_synthetic)"""
    legal = np.asarray(LEGAL[game], dtype=np.int32)
    env = np.arange(env_offset, env_offset + n, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ (env << np.uint64(32)) ^ np.uint64(t))
    return legal[(h % np.uint64(len(legal))).astype(np.int64)]
