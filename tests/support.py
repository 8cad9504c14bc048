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


def noop_count(noop_seed, global_env, episode_index, noop_max):
    """the engine's default no-op rule (include/toybox_amd.h, tbx_agent_init): 1 + splitmix64(seed ^ env << 32 ^ k) % noop_max"""
    return 1 + int(splitmix64(int(noop_seed) ^ (int(global_env) << 32) ^ int(episode_index)) % np.uint64(noop_max))


# ---------------------------------------------------------------- Amidar state edits of the wrapper corner cases
# Shared by tests/golden/make_wrapper_golden.py (applied through the reference env's write_state_json) and by
# tests/test_preproc.py (applied to the fused engine): inputs of the fixtures, not expected outputs.

def amidar_edit_last_lives(js, lives, jump_timer, perimeter_from_start):
    """every enemy parked on the player, a jump that runs out `jump_timer` frames from now: the life goes when the jump ends;
    with perimeter_from_start the enemies respawn ON the player's start tile and the next frame costs another life"""
    js["lives"], js["jump_timer"] = lives, jump_timer
    for en in js["enemies"]:
        en["position"] = dict(js["player"]["position"])
        en["step"] = None
        if perimeter_from_start:
            en["ai"] = {"EnemyPerimeterAI": {"start": {"tx": 31, "ty": 15}}}
    return js


def read_buffer(engine, which, shape, dtype=np.uint8):
    """host copy of an engine-owned buffer (TBX_BUF_*): device memory of the HIP library, plain memory of the CPU checker"""
    import ctypes as C
    ptr, nbytes = engine.device_buffer(which)
    out = np.empty(shape, dtype)
    assert out.nbytes == nbytes, (out.nbytes, nbytes)
    if hasattr(engine._lib, "orc_splitmix64"):
        C.memmove(out.ctypes.data, ptr, nbytes)
    else:
        from toybox_amd import hip
        engine.sync()
        hip.memcpy_dtoh(out, ptr, nbytes)
    return out


def stack_from_ring(ring, head):
    """uint8[stack][N][h][w] + the newest slot -> uint8[N][h][w][stack], oldest first (include/toybox_amd.h, new_plane = 2)"""
    k = ring.shape[0]
    return np.stack([ring[(head + 1 + c) % k] for c in range(k)], axis=-1)
