"""HIP code paths that the main parity file does not reach, each held to the CPU oracle bit for bit:

* tbx_apply_input (Toybox.apply_action(Input), /root/reference/scripts/utils/test_games.py:13) on every game, mixed with
  batch steps -- for Breakout this is the wave-per-env kernel brk_step_kernel<false>;
* the whole batch protocol with TBX_OPT_STEP_FORM = 2 (wave-per-env Breakout step instead of thread-per-env);
* the several-waves-per-frame rasteriser launches at other split factors than the default (TBX_OPT_RENDER_SPLIT);
* the pipelined mode (TBX_OPT_PIPELINE: steps beside renders, overlapped renders, double-buffered outputs and frames);
* BASELINE config 5's per-GPU share: three games x 10 922 envs on three HIP streams (toybox_amd.parallel.MixedBatch);
* every TBX_BUF_* id of tbx_device_buffer on both libraries;
* the cross-stream ordering rule of the C-ABI (async "_device" calls on the NULL stream, then host-pointer calls).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from support import LEGAL, synthetic_actions
from toybox_amd import Engine, ToyboxAmdError, _abi

GAMES = ["breakout", "space_invaders", "amidar", "gridworld"]

BUTTONS = {0: 0, 1: 16, 2: 4, 3: 2, 4: 1, 5: 8, 11: 2 | 16, 12: 1 | 16}   # ALE id -> TBX_BTN_* mask (constants.py:16-35)


def _pair(game, n, hip_lib, oracle_lib, seed=1234):
    g, o = Engine(game, n, lib=hip_lib), Engine(game, n, lib=oracle_lib)
    for e in (g, o):
        e.seed(seed)
        e.new_game()
    return g, o


def _same_states(g, o, envs, what=""):
    for i in envs:
        assert bytes(g.get_state(int(i))) == bytes(o.get_state(int(i))), "%s env %d" % (what, i)


@pytest.mark.gpu
@pytest.mark.parametrize("game", GAMES)
def test_apply_input_rollout_parity(game, hip_lib, oracle_lib):
    """One env at a time through tbx_apply_input (raw button masks, including BUTTON2 and opposing directions), with a
    batch step every 16th frame: states, the step outputs the call leaves in TBX_BUF_*, and frames equal the oracle."""
    n, frames = 6, 700
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=31)
    rng = np.random.default_rng(7)
    for t in range(frames):
        if t % 16 == 15:
            a = synthetic_actions(game, n, t)
            rg, ro = g.step(a), o.step(a)
            for x, y in zip(rg, ro):
                assert np.array_equal(x, y), t
            over = rg[1].astype(np.uint8)
            if over.any():
                g.new_game(over)
                o.new_game(over)
        else:
            a = synthetic_actions(game, n, t, seed=99)
            for i in range(n):
                b = BUTTONS[int(a[i])]
                if rng.random() < 0.1:
                    b = int(rng.integers(0, 64))          # any mask the Input struct can express
                g.apply_input(i, b)
                o.apply_input(i, b)
        if t % 50 == 49 or t < 3:
            _same_states(g, o, range(n), "frame %d" % t)
            assert np.array_equal(g.render(3), o.render(3))
            for x, y in zip(g.scalars(), o.scalars()):
                assert np.array_equal(x, y)
    _same_states(g, o, range(n), "end")


@pytest.mark.gpu
def test_apply_input_many_envs_long(hip_lib, oracle_lib):
    """Breakout's wave-per-env step over a long horizon (lives lost, new balls from the RNG, levels): every env of a
    64-env engine driven only through tbx_apply_input."""
    n = 64
    g, o = _pair("breakout", n, hip_lib, oracle_lib, seed=5)
    for t in range(1500):
        a = synthetic_actions("breakout", n, t)
        for i in range(n):
            g.apply_input(i, BUTTONS[int(a[i])])
            o.apply_input(i, BUTTONS[int(a[i])])
        if t % 300 == 299:
            _same_states(g, o, range(n), "frame %d" % t)
            over = g.scalars()[3].astype(np.uint8)
            assert np.array_equal(over, o.scalars()[3].astype(np.uint8))
            if over.any():
                g.new_game(over)
                o.new_game(over)
    lives = g.scalars()[1]
    assert (lives < 5).any(), "no life was ever lost: the reset-ball path was not reached"


_SUB = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from toybox_amd import Engine, _abi, _lib
from support import synthetic_actions
orc = ctypes.CDLL(os.path.join({root!r}, "oracle", "liboracle.so")); _abi.bind(orc)
hip = _lib.load()
{body}
print("SUB_OK")
"""


def _run_sub(body, env, timeout=900):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _SUB.format(root=ROOT, body=body)], env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0 and "SUB_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_breakout_wave_per_env_step_kernel_parity(oracle_lib):
    """TBX_OPT_STEP_FORM = 2: the batch protocol and the agent pipeline run on brk_step_kernel<false> (one wavefront per env)
    instead of the thread-per-env kernel; same bar as test_rollout_parity."""
    body = r"""
n, steps = 2048, 900
g, o = Engine("breakout", n, lib=hip), Engine("breakout", n, lib=orc)
for e in (g, o):
    e.set_option(_abi.OPT_STEP_FORM, _abi.STEP_FORM_WAVE_PER_ENV)
    e.seed(1234); e.new_game()
done = 0
for t in range(steps):
    a = synthetic_actions("breakout", n, t)
    rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)
    for x, y in zip(rg, ro):
        assert np.array_equal(x, y), t
    done += int(rg[1].sum())
    if t % 300 == 299:
        assert np.array_equal(g.render(3)[::64], o.render(3)[::64])
for i in range(n):
    assert bytes(g.get_state(i)) == bytes(o.get_state(i)), i
assert done > 0
# device-generated actions and the agent pipeline on the same kernel
for t in range(200):
    g.step_synthetic(1337, t, env_offset=3, auto_reset=True)
    o.step(synthetic_actions("breakout", n, t, seed=1337, env_offset=3), auto_reset=True)
g.sync()
for i in range(0, n, 5):
    assert bytes(g.get_state(i)) == bytes(o.get_state(i)), i
for e in (g, o):
    e.agent_init(skip=4, episodic_life=True, fire_reset=True, noop_max=8, noop_seed=3)
assert np.array_equal(g.agent_reset(), o.agent_reset())
for t in range(120):
    a = synthetic_actions("breakout", n, t, seed=11)
    xg, xo = g.agent_step(a), o.agent_step(a)
    for x, y in zip(xg, xo):
        assert np.array_equal(x, y), t
"""
    _run_sub(body, {})


@pytest.mark.gpu
@pytest.mark.parametrize("form", [1, 2])
def test_amidar_step_kernel_forms_parity(form, oracle_lib):
    """Amidar's batch step has two forms, chosen by batch size (thread per env from 32 768 envs up, wavefront per env
    below): TBX_OPT_STEP_FORM forces either one at a batch size the oracle finishes, through the batch protocol, device
    actions and the agent pipeline with every reset-time wrapper on."""
    body = r"""
n, steps = 1100, 700      # not a multiple of 64: the thread form's last wave is ragged
g, o = Engine("amidar", n, lib=hip), Engine("amidar", n, lib=orc)
for e in (g, o):
    e.set_option(_abi.OPT_STEP_FORM, @FORM@)
    e.seed(77); e.new_game()
for t in range(steps):
    a = synthetic_actions("amidar", n, t)
    rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)
    for x, y in zip(rg, ro):
        assert np.array_equal(x, y), t
    if t % 233 == 232:
        assert np.array_equal(g.render(3)[::37], o.render(3)[::37])
for i in range(n):
    assert bytes(g.get_state(i)) == bytes(o.get_state(i)), i
for t in range(100):
    g.step_synthetic(1337, t, env_offset=3, auto_reset=True)
    o.step(synthetic_actions("amidar", n, t, seed=1337, env_offset=3), auto_reset=True)
g.sync()
for i in range(0, n, 3):
    assert bytes(g.get_state(i)) == bytes(o.get_state(i)), i
for e in (g, o):
    e.agent_init(skip=4, episodic_life=True, fire_reset=True, noop_max=8, noop_seed=3)
assert np.array_equal(g.agent_reset(), o.agent_reset())
for t in range(150):
    a = synthetic_actions("amidar", n, t, seed=11)
    xg, xo = g.agent_step(a), o.agent_step(a)
    for x, y in zip(xg, xo):
        assert np.array_equal(x, y), t
for i in range(0, n, 3):
    assert bytes(g.get_state(i)) == bytes(o.get_state(i)), i
"""
    _run_sub(body.replace("@FORM@", str(form)), {})


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 2, 3, 5, 7, 16])
def test_render_split_factors_parity(split, oracle_lib):
    """The rasterisers take `split` waves per frame (launch-time choice, by default a function of game, channels and batch
    size).  Every factor must paint the same bytes: all games, all channel counts, batch sizes that do not fill a block."""
    body = r"""
for game in ("breakout", "space_invaders", "amidar", "gridworld"):
    for n in (3, 130):
        g, o = Engine(game, n, lib=hip), Engine(game, n, lib=orc)
        for e in (g, o):
            e.set_option(_abi.OPT_RENDER_SPLIT, @SPLIT@)
            e.seed(9); e.new_game()
        for t in range(260):
            a = synthetic_actions(game, n, t)
            g.step(a, auto_reset=True); o.step(a, auto_reset=True)
            if t in (0, 129, 259):
                for c in (1, 3, 4):
                    assert np.array_equal(g.render(c), o.render(c)), (game, n, t, c)
        assert np.array_equal(g.render_env(n - 1, 3), o.render_env(n - 1, 3))
"""
    _run_sub(body.replace("@SPLIT@", str(split)), {})


@pytest.mark.gpu
def test_mixed_batch_config5_share_on_three_streams(hip_lib, oracle_lib):
    """BASELINE config 5, one GPU's share: Breakout + Amidar + SpaceInvaders, 10 922 envs each, stepped with in-kernel
    actions by GLOBAL env index and rendered on three HIP streams -- equal to three oracle engines fed the same rule:
    per-step outputs (through the packed gather record), full states of a sample, frames of a sample."""
    from toybox_amd import hip
    from toybox_amd.parallel import MixedBatch, unpack_records
    games, per, steps, off = ["breakout", "amidar", "space_invaders"], 10922, 150, 32768
    old = os.environ.get("TBX_ORACLE_THREADS")
    os.environ["TBX_ORACLE_THREADS"] = str(min(16, os.cpu_count() or 1))
    try:
        mb = MixedBatch(games, per, engine_factory=lambda g, n: Engine(g, n, lib=hip_lib), global_offset=off)
        ref = MixedBatch(games, per, engine_factory=lambda g, n: Engine(g, n, lib=oracle_lib), global_offset=off)
    finally:
        if old is None:
            os.environ.pop("TBX_ORACLE_THREADS")
        else:
            os.environ["TBX_ORACLE_THREADS"] = old
    streams = [hip.Stream() for _ in games]
    mb.attach_streams([s.ptr for s in streams])
    rec = np.empty(per, np.uint64)
    n_done = 0
    for t in range(steps):
        if t < steps // 2:
            mb.step_synthetic(1337, t)
            mb.render_device(3)
        else:                                   # the fused rollout call per segment (one launch for Breakout)
            mb.render_step_synthetic(1337, t, 3)
        ref.step_synthetic(1337, t)
        if t % 10 == 9 or t == steps - 1:
            for s in streams:
                s.synchronize()
            for ge, oe in zip(mb.engines, ref.engines):
                p, nbytes = ge.device_buffer(_abi.BUF_PACKED)
                assert nbytes == 8 * per
                hip.memcpy_dtoh(rec, p, nbytes)
                q, _ = oe.device_buffer(_abi.BUF_PACKED)
                want = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint64)), (per,))
                assert np.array_equal(rec, want), (ge.game, t)
                n_done += int(unpack_records(rec)[1].sum())
    mb.render_device(3)                          # (a fused call's frame shows the state before its step)
    for s in streams:
        s.synchronize()
    sample = list(range(0, per, 97)) + [per - 1]
    one = None
    for ge, oe in zip(mb.engines, ref.engines):
        for x, y in zip(ge.scalars(), oe.scalars()):
            assert np.array_equal(x, y), ge.game
        _same_states(ge, oe, sample, ge.game)
        # frames of the last render_device, straight out of the engine-owned frame batch
        p, nbytes = ge.device_buffer(_abi.BUF_FRAME)
        fb = ge.height * ge.width * 3
        assert nbytes >= per * fb
        one = np.empty(fb, np.uint8)
        for i in sample[::8]:
            hip.memcpy_dtoh(one, p + i * fb, fb)
            assert np.array_equal(one.reshape(ge.height, ge.width, 3), oe.render_env(i, 3)), (ge.game, i)
    assert mb.n_envs == 3 * per
    mb.close()
    ref.close()
    for s in streams:
        s.close()


ALL_BUFS = ["BUF_REWARD", "BUF_DONE", "BUF_LIVES", "BUF_SCORE", "BUF_FRAME", "BUF_PACKED", "BUF_AGENT_OBS", "BUF_AGENT_REWARD",
            "BUF_AGENT_DONE", "BUF_AGENT_EP_DONE", "BUF_AGENT_EP_RETURN", "BUF_AGENT_EP_LENGTH", "BUF_GATHERED", "BUF_AGENT_PLANE", "BUF_AGENT_RING",
            "BUF_ROLLOUT_FRAMES", "BUF_ROLLOUT_PACKED"]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request, oracle_lib):
    if request.param == "oracle":
        return oracle_lib
    from toybox_amd import _lib
    return _lib.load()


def test_every_device_buffer_id(lib):
    """Every TBX_BUF_* the header declares is addressable once its producer ran, with the documented size; agent buffers
    before tbx_agent_init and unknown ids are TBX_E_INVALID on both libraries."""
    import re
    text = open(os.path.join(ROOT, "include", "toybox_amd.h")).read()
    declared = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+TBX_(BUF_[A-Z_]+)\s+(\d+)", text)}
    assert sorted(declared) == sorted(ALL_BUFS)
    for name, v in declared.items():
        assert getattr(_abi, name) == v
    n = 5
    e = Engine("breakout", n, lib=lib)
    for name in ALL_BUFS:
        if "AGENT" in name or "ROLLOUT" in name or name == "BUF_GATHERED":
            with pytest.raises(ToyboxAmdError) as ei:
                e.device_buffer(declared[name])
            assert ei.value.code == _abi.E_INVALID, name
    for bad in (-1, 17, 99):
        with pytest.raises(ToyboxAmdError) as ei:
            e.device_buffer(bad)
        assert ei.value.code == _abi.E_INVALID
    e.step([1] * n)
    e.render_device(0, 3)
    e.sync()
    e.agent_init(skip=2, out_h=42, out_w=60, stack=3)
    with pytest.raises(ToyboxAmdError) as ei:                # the newest plane alone exists only when asked for
        e.device_buffer(_abi.BUF_AGENT_PLANE)
    assert ei.value.code == _abi.E_INVALID
    e.agent_init(skip=2, out_h=42, out_w=60, stack=3, new_plane=True)
    e.agent_reset()
    e.agent_step([1] * n)
    e.gather_init(1, 0, e.gather_unique_id(), records_per_rank=n + 3)
    e.rollout_synthetic(7, 0, 2, channels=3)                 # (a collective per step: the two single calls; rows of N records)
    e.sync()
    want = {"BUF_ROLLOUT_FRAMES": 2 * n * 160 * 240 * 3, "BUF_ROLLOUT_PACKED": 2 * 8 * n, "BUF_GATHERED": 8 * (n + 3),"BUF_REWARD": 4 * n, "BUF_DONE": n, "BUF_LIVES": 4 * n, "BUF_SCORE": 4 * n, "BUF_FRAME": n * 160 * 240 * 3,
            "BUF_PACKED": 8 * n, "BUF_AGENT_OBS": n * 42 * 60 * 3, "BUF_AGENT_REWARD": 4 * n, "BUF_AGENT_DONE": n,
            "BUF_AGENT_EP_DONE": n, "BUF_AGENT_EP_RETURN": 4 * n, "BUF_AGENT_EP_LENGTH": 4 * n, "BUF_AGENT_PLANE": n * 42 * 60}
    seen = set()
    for name in ALL_BUFS:
        if name == "BUF_AGENT_RING":                         # the ring of planes exists INSTEAD of the stack (new_plane = 2)
            with pytest.raises(ToyboxAmdError) as ei:
                e.device_buffer(declared[name])
            assert ei.value.code == _abi.E_INVALID
            continue
        p, b = e.device_buffer(declared[name])
        assert p and b == want[name], (name, p, b)
        if name != "BUF_FRAME":                              # (after a chunk TBX_BUF_FRAME may name the chunk's last frame)
            seen.add(p)
    assert len(seen) == len(ALL_BUFS) - 2, "two buffer ids share an address"
    e.agent_init(skip=2, out_h=42, out_w=60, stack=3, new_plane=2)
    e.agent_reset()
    p, b = e.device_buffer(_abi.BUF_AGENT_RING)
    assert p and b == n * 42 * 60 * 3
    q, b = e.device_buffer(_abi.BUF_AGENT_PLANE)             # the ring's newest slot
    assert b == n * 42 * 60 and q == p + e.agent_ring_head() * b
    with pytest.raises(ToyboxAmdError) as ei:
        e.device_buffer(_abi.BUF_AGENT_OBS)
    assert ei.value.code == _abi.E_INVALID
    e.close()


@pytest.mark.gpu
def test_device_buffer_contents_match_host_outputs(hip_lib):
    """What the ids point at is what the host-pointer calls return."""
    from toybox_amd import hip
    n = 300
    e = Engine("amidar", n, lib=hip_lib)
    e.seed(3)
    e.new_game()
    for t in range(200):
        out = e.step(synthetic_actions("amidar", n, t), auto_reset=True)
    for which, host, dt in ((_abi.BUF_REWARD, out[0], np.int32), (_abi.BUF_DONE, out[1].astype(np.uint8), np.uint8),
                            (_abi.BUF_LIVES, out[2], np.int32), (_abi.BUF_SCORE, out[3], np.int32)):
        p, b = e.device_buffer(which)
        got = np.empty(n, dt)
        hip.memcpy_dtoh(got, p, b)
        assert np.array_equal(got, host), which
    e.agent_init(skip=4)
    e.agent_reset()
    for t in range(60):
        obs, rew, done = e.agent_step(synthetic_actions("amidar", n, t))
    ended, ret, length = e.agent_episodes()
    for which, host in ((_abi.BUF_AGENT_OBS, obs), (_abi.BUF_AGENT_REWARD, rew), (_abi.BUF_AGENT_DONE, done.astype(np.uint8)),
                        (_abi.BUF_AGENT_EP_DONE, ended.astype(np.uint8)), (_abi.BUF_AGENT_EP_RETURN, ret),
                        (_abi.BUF_AGENT_EP_LENGTH, length)):
        p, b = e.device_buffer(which)
        got = np.empty(host.shape, host.dtype)
        assert b == got.nbytes
        hip.memcpy_dtoh(got, p, b)
        assert np.array_equal(got, host), which
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["breakout", "space_invaders", "amidar"])
def test_null_stream_async_then_host_calls_are_ordered(game, hip_lib, oracle_lib):
    """The "_device" entry points run on the caller's stream (here the NULL stream, the Python default), the host-pointer
    ones on the engine's own non-blocking stream.  Calls on one handle take effect in program order without an explicit
    tbx_sync in between (include/toybox_amd.h, conventions)."""
    n = 20000
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=8)
    for rnd in range(6):
        for t in range(25):
            g.step_synthetic(1337, 25 * rnd + t, auto_reset=True)                  # async, NULL stream
            o.step(synthetic_actions(game, n, 25 * rnd + t, seed=1337), auto_reset=True)
        a = synthetic_actions(game, n, rnd, seed=4)
        rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)           # engine stream, no sync in between
        for x, y in zip(rg, ro):
            assert np.array_equal(x, y), rnd
        g.render_device(0, 1)                                                      # async again ...
        _same_states(g, o, range(0, n, 1999), "round %d" % rnd)                    # ... then a host-pointer read


def test_gather_one_rank_equals_local_records(lib):
    """tbx_gather over a one-rank communicator (RCCL on the GPU box): the gathered block is the engine's own packed
    records, padded to the slot width; a step queued after a gather does not overtake it; the max-reduction returns its
    argument.  The N > 1 case differs only in the communicator (same code path, same kernels)."""
    from toybox_amd.parallel import pack_records
    n, width = 1000, 1024
    e = Engine("breakout", n, lib=lib)
    e.seed(77)
    e.new_game()
    with pytest.raises(ToyboxAmdError):
        e.gather()
    e.gather_init(1, 0, e.gather_unique_id(), records_per_rank=width)
    with pytest.raises(ToyboxAmdError):
        e.gather_init(1, 0, e.gather_unique_id(), records_per_rank=n - 1)
    for t in range(120):
        a = synthetic_actions("breakout", n, t)
        reward, done, lives, _ = e.step(a, auto_reset=True)
        e.gather()
        if t % 3 == 0:
            e.step_synthetic(5, t, auto_reset=True)       # queued right behind the gather: must not change what it sends
        got = e.gather_host()
        assert got.shape == (1, width)
        assert np.array_equal(got[0, :n], pack_records(reward, done, lives)), t
        assert not got[0, n:].any()
    assert e.gather_reduce_max(3.25) == 3.25
    e.close()


@pytest.mark.parametrize("game,K", [("breakout", 4), ("space_invaders", 3), ("amidar", 5), ("gridworld", 2)])
def test_gather_ring_sends_k_steps_with_one_collective(game, K, lib, oracle_lib):
    """TBX_OPT_GATHER_EVERY = K ("per step or per K steps"): the step kernels write their records into slot j of a
    ring [K][width], tbx_gather only counts, the K-th call sends the ring with ONE collective; gathered layout [1][K][width],
    slot j = the records of the j-th step since the last collective.  TBX_BUF_PACKED names the slot of the most recent step;
    steps queued behind a collective (the other ring fills meanwhile) do not change what it sends; a step without a tbx_gather
    rewrites its slot; leaving ring mode (a new tbx_gather_init with K = 1) hands the records back to the engine's own array."""
    from toybox_amd.parallel import pack_records
    n, width = 333, 340
    e, ref = Engine(game, n, lib=lib), Engine(game, n, lib=oracle_lib)
    for x in (e, ref):
        x.seed(21)
        x.new_game()
    e.set_option(_abi.OPT_GATHER_EVERY, K)
    with pytest.raises(ToyboxAmdError):
        e.set_option(_abi.OPT_GATHER_EVERY, 0)
    e.gather_init(1, 0, e.gather_unique_id(), records_per_rank=width)
    assert e.gather_every() == K and e.gather_fill() == 0
    assert e.device_buffer(_abi.BUF_GATHERED)[1] == 8 * K * width
    assert e.get_option(_abi.OPT_PIPELINE_ACTIVE) == 0
    window, slots, t = [], set(), 0
    for rnd in range(7):
        for j in range(K):
            if rnd == 3 and j == 1:                         # a step nobody gathers: the next one takes its slot
                e.step_synthetic(1337, t, auto_reset=True)
                ref.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
                t += 1
            e.step_synthetic(1337, t, auto_reset=True)
            r, d, l, _ = ref.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
            t += 1
            window.append(pack_records(r, d, l))
            slots.add(e.device_buffer(_abi.BUF_PACKED)[0])
            e.gather()
            assert e.gather_fill() == (j + 1) % K
        if rnd % 2:                                         # two steps of the NEXT ring queued before the result is read
            for _ in range(2):
                e.step_synthetic(1337, t, auto_reset=True)
                ref.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
                t += 1
        got = e.gather_host()
        assert got.shape == (1, K, width)
        for j in range(K):
            assert np.array_equal(got[0, j, :n], window[j]), (rnd, j)
            assert not got[0, j, n:].any()
        window = []
    assert len(slots) == 2 * K                              # two rings of K slots
    e.sync()
    for i in range(0, n, 37):
        assert bytes(e.get_state(i)) == bytes(ref.get_state(i))
    e.set_option(_abi.OPT_GATHER_EVERY, 1)
    e.gather_init(1, 0, e.gather_unique_id(), records_per_rank=width)
    assert e.gather_every() == 1 and e.device_buffer(_abi.BUF_PACKED)[0] not in slots
    r, d, l, _ = e.step(synthetic_actions(game, n, t), auto_reset=True)
    e.gather()
    got = e.gather_host()
    assert got.shape == (1, width) and np.array_equal(got[0, :n], pack_records(r, d, l))
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 4])
def test_gather_overlaps_with_render_and_keeps_parity(K, hip_lib, oracle_lib):
    """The bench loop shape: step -> gather -> render on the caller's stream, 8 192 envs (the 8-GPU strong-scaling share):
    gathered records and final states equal the oracle's."""
    from toybox_amd import hip
    from toybox_amd.parallel import unpack_records
    n = 8192
    g, o = _pair("breakout", n, hip_lib, oracle_lib)
    g.set_option(_abi.OPT_GATHER_EVERY, K)
    g.gather_init(1, 0, g.gather_unique_id())
    st = hip.Stream()
    last = []
    for t in range(300):
        g.step_synthetic(1337, t, auto_reset=True, stream=st.ptr)
        g.gather(stream=st.ptr)
        g.render_device(0, 3, stream=st.ptr)
        r, d, l, _ = o.step(synthetic_actions("breakout", n, t), auto_reset=True)
        last = (last + [(r, d, l)])[-K:]
        if (t + 1) % K == 0 and (t % 7 < K or t > 290):     # (K > 1: a collective has just gone out, with the last K steps)
            got = g.gather_host()[0].reshape(K, n)
            for j in range(K):
                rr, dd, ll = unpack_records(got[j])
                r, d, l = last[j]
                assert np.array_equal(rr, r) and np.array_equal(dd, d) and np.array_equal(ll, np.clip(l, 0, 255)), (t, j)
    st.synchronize()
    _same_states(g, o, range(0, n, 211), "end")
    st.close()


@pytest.mark.parametrize("game", GAMES)
def test_step1_equals_batch_step(game, lib, oracle_lib):
    """tbx_step1 (one frame of one env by ALE id, outputs included) == tbx_step on a twin engine -- on a one-env engine of
    the HIP library this is the resident step kernel (mailbox in pinned memory, no launch per frame), interleaved here with
    calls that stop it (frames, state reads and writes, new games), idle pauses that let it leave, and an illegal action."""
    import time
    a1, ref = Engine(game, 1, lib=lib), Engine(game, 1, lib=oracle_lib)
    for e in (a1, ref):
        e.seed(21)
        e.new_game()
    rng = np.random.default_rng(3)
    for t in range(2500):
        a = int(synthetic_actions(game, 1, t)[0])
        r, d, l, s = a1.step1(0, a, auto_reset=True)
        rr, dd, ll, ss = ref.step([a], auto_reset=True)
        assert (r, d, l, s) == (int(rr[0]), bool(dd[0]), int(ll[0]), int(ss[0])), t
        k = rng.random()
        if k < 0.01:
            assert np.array_equal(a1.render_env(0, 3), ref.render_env(0, 3)), t
        elif k < 0.02:
            assert bytes(a1.get_state(0)) == bytes(ref.get_state(0)), t
        elif k < 0.025:
            st = ref.get_state(0)
            a1.set_state(0, st)
            ref.set_state(0, st)
        elif k < 0.03:
            a1.new_game()
            ref.new_game()
        elif k < 0.032:
            time.sleep(0.08)                      # longer than the resident kernel's idle time: it leaves, the next call restarts it
    assert bytes(a1.get_state(0)) == bytes(ref.get_state(0))
    with pytest.raises(ToyboxAmdError) as ei:
        a1.step1(0, 99)
    assert ei.value.code == _abi.E_ACTION
    a1.step1(0, 0)
    ref.step([0])
    ref.step([0])
    assert bytes(a1.get_state(0)) == bytes(ref.get_state(0))
    a1.close()


def test_step1_on_a_batch_engine(lib, oracle_lib):
    n = 5
    e, ref = Engine("breakout", n, lib=lib), Engine("breakout", n, lib=oracle_lib)
    for x in (e, ref):
        x.seed(2)
        x.new_game()
    for t in range(200):
        env = t % n
        a = int(synthetic_actions("breakout", n, t)[env])
        out = e.step1(env, a)
        ref.apply_input(env, BUTTONS[a])
        sc, lv, _, _ = ref.scalars()
        assert out[2] == int(lv[env]) and out[3] == int(sc[env])
    for i in range(n):
        assert bytes(e.get_state(i)) == bytes(ref.get_state(i))


@pytest.mark.gpu
@pytest.mark.parametrize("game,n,mode", [("breakout", 40000, 2), ("breakout", 40000, 3), ("breakout", 3000, 3), ("breakout", 3000, 1),
                                         ("space_invaders", 20000, 2), ("space_invaders", 3000, 3), ("amidar", 20000, 3), ("amidar", 3000, 2),
                                         ("amidar", 3000, 3), ("amidar", 12000, 2), ("gridworld", 5000, 3)])
@pytest.mark.parametrize("same_stream", [True, False])
def test_pipelined_mode_keeps_program_order(same_stream, game, n, mode, hip_lib, oracle_lib):
    """TBX_OPT_PIPELINE: tbx_step_synthetic on Breakout runs on the engine's step stream BESIDE the rasteriser launch queued
    before it (two buffers of records and of step outputs); with value 3 consecutive rasteriser launches alternate between two
    internal streams and two frame buffers; SpaceInvaders likewise (records), Amidar through the record-prep kernel that follows
    its step on the step's stream (GridWorld, which ignores the option, walks the same call pattern as the control).
    What the caller sees must stay program order: every frame is the frame of the step before it, device buffers read behind
    the caller's stream are the step's, and calls of every other kind in between (state reads and writes, new games,
    host-pointer steps and renders, a device-action step, a second render of one frame, two steps in a row) join the
    pipeline.  Bench pattern; 40 000 envs so that a render launch is long enough to be overtaken."""
    from toybox_amd import hip
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=31)
    g.set_option(_abi.OPT_PIPELINE, mode)
    assert g.get_option(_abi.OPT_PIPELINE) == mode
    s_step, s_render = hip.Stream(), hip.Stream()
    sp, rp = (s_render.ptr, s_render.ptr) if same_stream else (s_step.ptr, s_render.ptr)
    H, W = g.height, g.width
    one = np.empty((H, W, 3), np.uint8)
    packed = np.empty(n, np.uint64)
    sample = list(range(0, n, 997)) + [n - 1]
    rng = np.random.default_rng(2)
    t = 0

    def check_frame(tag):
        hip.synchronize()
        p, nbytes = g.device_buffer(_abi.BUF_FRAME)
        for i in sample:
            hip.memcpy_dtoh(one, p + i * H * W * 3, H * W * 3)
            assert np.array_equal(one, o.render_env(i, 3)), (tag, i)

    for rnd in range(12):
        for k in range(int(rng.integers(3, 30))):           # the bench loop: step, render, step, render ...
            g.step_synthetic(1337, t, auto_reset=True, stream=sp)
            g.render_device(0, 3, stream=rp)
            o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
            t += 1
        check_frame("round %d" % rnd)
        # the step's outputs, read behind the stream the step call named
        g.step_synthetic(1337, t, auto_reset=True, stream=sp)
        ro = o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
        t += 1
        (s_render if same_stream else s_step).synchronize()
        p, nbytes = g.device_buffer(_abi.BUF_PACKED)
        hip.memcpy_dtoh(packed, p, 8 * n)
        from toybox_amd.parallel import pack_records
        assert np.array_equal(packed, pack_records(ro[0], ro[1], ro[2])), rnd
        kind = rnd % 6
        if kind == 0:                                       # state read right behind a step that ran ahead of a render
            g.render_device(0, 3, stream=rp)
            g.step_synthetic(1337, t, auto_reset=True, stream=sp)
            o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
            t += 1
            _same_states(g, o, sample, "read %d" % rnd)
        elif kind == 1:                                     # state write, then straight back into the loop
            st = o.get_state(5)
            for e in (g, o):
                e.set_state(11, st)
        elif kind == 2:                                     # masked new game
            m = (np.arange(n) % 7 == 0).astype(np.uint8)
            g.new_game(m); o.new_game(m)
        elif kind == 3:                                     # host-pointer step and render
            a = synthetic_actions(game, n, t, seed=5)
            rg, ro = g.step(a, auto_reset=True), o.step(a, auto_reset=True)
            for x, y in zip(rg, ro):
                assert np.array_equal(x, y)
            for i in sample[:8]:
                assert np.array_equal(g.render_env(i, 3), o.render_env(i, 3))
        elif kind == 4:                                     # two renders of one frame on two streams, two steps in a row
            g.render_device(0, 3, stream=rp)
            g.render_device(0, 3, stream=sp)
            for _ in range(2):
                g.step_synthetic(1337, t, auto_reset=True, stream=sp)
                o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
                t += 1
        else:                                               # a render, then a step that must NOT overtake a state read
            g.render_device(0, 1, stream=rp)
            _same_states(g, o, sample[:5], "gray %d" % rnd)
    g.render_device(0, 3, stream=rp)
    check_frame("end")
    _same_states(g, o, sample, "end")
    for x, y in zip(g.scalars(), o.scalars()):
        assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_results_stay_valid_for_readers_queued_before_the_next_call(mode, hip_lib, oracle_lib):
    """The stream-ordering contract of the TBX_BUF_* outputs, with and without the pipelined mode: device-side copies of
    TBX_BUF_PACKED queued on the caller's stream right behind a step AND behind the render that follows it, and copies of frames
    queued behind the render -- nothing synchronised until the very end, so in pipelined mode the next step (beside the render)
    and the next render (on the other internal stream) are in flight while the copies still wait their turn."""
    from toybox_amd import hip
    from toybox_amd.parallel import pack_records
    n, T, game = 40000, 16, "breakout"
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=8)
    g.set_option(_abi.OPT_PIPELINE, mode)
    H, W = g.height, g.width
    fb = H * W * 3
    sample = [0, 1, n // 2, n - 1]
    s = hip.Stream()
    c1, c2, cf = hip.malloc(8 * n * T), hip.malloc(8 * n * T), hip.malloc(fb * len(sample) * T)
    want = []
    for t in range(T):
        g.step_synthetic(1337, t, auto_reset=True, stream=s.ptr)
        p, _ = g.device_buffer(_abi.BUF_PACKED)
        hip.memcpy_dtod_async(c1 + 8 * n * t, p, 8 * n, s)
        g.render_device(0, 3, stream=s.ptr)
        assert g.device_buffer(_abi.BUF_PACKED)[0] == p          # a render does not move the step outputs
        hip.memcpy_dtod_async(c2 + 8 * n * t, p, 8 * n, s)
        f, _ = g.device_buffer(_abi.BUF_FRAME)
        for k, i in enumerate(sample):
            hip.memcpy_dtod_async(cf + fb * (len(sample) * t + k), f + fb * i, fb, s)
        ro = o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
        want.append((pack_records(ro[0], ro[1], ro[2]), [o.render_env(i, 3) for i in sample]))
    s.synchronize()
    got, one = np.empty(n, np.uint64), np.empty((H, W, 3), np.uint8)
    for t in range(T):
        for c in (c1, c2):
            hip.memcpy_dtoh(got, c + 8 * n * t, 8 * n)
            assert np.array_equal(got, want[t][0]), (mode, t)
        for k in range(len(sample)):
            hip.memcpy_dtoh(one, cf + fb * (len(sample) * t + k), fb)
            assert np.array_equal(one, want[t][1][k]), (mode, t, k)
    for c in (c1, c2, cf):
        hip.free(c)
    g.sync()
    _same_states(g, o, sample, "end")


@pytest.mark.gpu
@pytest.mark.parametrize("game,n", [("breakout", 40000), ("space_invaders", 24000), ("amidar", 30000), ("amidar", 9000)])
@pytest.mark.parametrize("mode", [2, 3])
def test_pipelined_render_of_a_rewritten_state_is_not_overtaken_by_the_next_step(mode, game, n, hip_lib, oracle_lib):
    """ADVICE r03 (engine.hip, pipe_step): after new_game / set_state the render records are stale, so the first render of the
    pipelined loop starts from LIVE state (the record-prep kernel, or Amidar's state-reading rasteriser) on the render stream;
    the tbx_step_synthetic that follows runs on an internal stream and rewrites that state -- it has to wait for that reader.
    [new_game | set_state -> render_device -> step_synthetic x 3] with device-side copies of sampled frames queued straight
    behind the render and nothing synchronised in between: every copy must show the state BEFORE the steps."""
    from toybox_amd import hip
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=77)
    g.set_option(_abi.OPT_PIPELINE, mode)
    H, W = g.height, g.width
    fb = H * W * 3
    sample = [0, 1, n // 3, n // 2, n - 2, n - 1]
    s = hip.Stream()
    rounds = 10
    hold = hip.malloc(fb * len(sample) * rounds)
    want = []
    t = 0
    for rnd in range(rounds):
        for k in range(3 + rnd % 3):                         # the loop proper, so that records exist and the lanes are busy
            g.step_synthetic(1337, t, auto_reset=True, stream=s.ptr)
            g.render_device(0, 3, stream=s.ptr)
            o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
            t += 1
        if rnd % 2 == 0:                                     # every env starts over ...
            m = np.ones(n, np.uint8)
            m[1] = 0
            g.new_game(m); o.new_game(m)
        else:                                                # ... or sampled envs take another env's state
            for i in sample:
                st = o.get_state((i * 7 + 3) % n)
                g.set_state(i, st); o.set_state(i, st)
        g.render_device(0, 3, stream=s.ptr)                  # records are stale: this launch reads live state
        f, _ = g.device_buffer(_abi.BUF_FRAME)
        for k, i in enumerate(sample):
            hip.memcpy_dtod_async(hold + fb * (len(sample) * rnd + k), f + fb * i, fb, s)
        want.append([o.render_env(i, 3) for i in sample])
        for _ in range(3):                                   # steps right behind it; they must not reach into that frame
            g.step_synthetic(1337, t, auto_reset=True, stream=s.ptr)
            o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
            t += 1
    s.synchronize()
    one = np.empty((H, W, 3), np.uint8)
    for rnd in range(rounds):
        for k, i in enumerate(sample):
            hip.memcpy_dtoh(one, hold + fb * (len(sample) * rnd + k), fb)
            assert np.array_equal(one, want[rnd][k]), (game, mode, rnd, i)
    hip.free(hold)
    g.sync()
    _same_states(g, o, sample, "end")


@pytest.mark.parametrize("game", GAMES)
def test_render_step_synthetic_call_contract(game, lib):
    """one call == tbx_render_device + tbx_step_synthetic on either library (host-visible results only: small batch)"""
    n = 9
    a, b = Engine(game, n, lib=lib), Engine(game, n, lib=lib)
    for e in (a, b):
        e.seed(4)
        e.new_game()
    for t in range(40):
        if t == 20 and game == "breakout":                  # intervention-written bricks: the call is two launches from here on
            for e in (a, b):
                st = e.get_state(2)
                st.bricks[7].col = 3
                st.bricks[30].x += 2.0
                e.set_state(2, st)
            assert a.get_option(_abi.OPT_RENDER_STEP_FUSED) == 0
        a.render_step_synthetic(1337, t, channels=3, auto_reset=True)
        fa = a.device_buffer(_abi.BUF_FRAME)
        b.render_device(0, 3)
        b.step_synthetic(1337, t, auto_reset=True)
        assert fa[1] == b.device_buffer(_abi.BUF_FRAME)[1]
    a.sync(); b.sync()
    for i in range(n):
        assert bytes(a.get_state(i)) == bytes(b.get_state(i))
        assert np.array_equal(a.render_env(i, 3), b.render_env(i, 3))
    with pytest.raises(ToyboxAmdError):
        a.render_step_synthetic(1337, 0, channels=2)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game,n,channels,overlap", [("breakout", 20000, 3, 2), ("breakout", 20000, 3, 1), ("breakout", 4096, 4, 0), ("breakout", 8192, 4, 2),
                                                     ("breakout", 300, 3, 1), ("breakout", 300, 3, 2), ("breakout", 5000, 1, 1),
                                                     ("space_invaders", 3000, 3, 0), ("space_invaders", 1500, 4, 1), ("space_invaders", 9000, 3, 0),
                                                     ("space_invaders", 17000, 3, 0), ("amidar", 2000, 3, 1), ("gridworld", 1000, 3, 0)])
def test_render_step_synthetic_equals_render_then_step(game, n, channels, overlap, hip_lib, oracle_lib):
    """tbx_render_step_synthetic = tbx_render_device followed by tbx_step_synthetic, bit for bit: the frame shows the state
    before the step, outputs and state are the step's.  For Breakout RGB / RGBA that is ONE launch (brk_render_step_kernel_w5:
    step blocks in front of the rasteriser's, the other records buffer); everywhere else two launches.  Calls of other kinds
    in between (state writes that invalidate the records, new games, host steps, a K-step gather ring, plain step / render
    pairs, the pipelined mode switched on) must not disturb it.  Frames are copied device-side right behind each call.
    overlap = TBX_OPT_FUSED_OVERLAP: 1 = consecutive fused launches on two lanes behind the device-side ticket (round 6), 2 =
    stream order, 0 = the engine's choice; engines without a fused launch ignore it."""
    from toybox_amd import hip
    from toybox_amd.parallel import pack_records
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=5)
    g.set_option(_abi.OPT_FUSED_OVERLAP, overlap)
    fused = game == "breakout" and channels >= 3
    if channels >= 3:                                       # (the read-only option answers for RGB / RGBA frames)
        assert g.get_option(_abi.OPT_FUSED_OVERLAP_ACTIVE) == (1 if fused and (overlap == 1 or (overlap == 0 and n <= 4096)) else 0)
    H, W = g.height, g.width
    fb = H * W * channels
    sample = sorted({0, 1, 255, 256, n // 2, n - 1} & set(range(n)))
    s = hip.Stream()
    T = 90
    hold = hip.malloc(fb * len(sample) * T)
    want, packed = [], np.empty(n, np.uint64)
    g.set_option(_abi.OPT_GATHER_EVERY, 3)
    g.gather_init(1, 0, g.gather_unique_id())
    for t in range(T):
        if t == 20:
            st = o.get_state(3)
            g.set_state(0, st); o.set_state(0, st)           # records stale: the fused launch rebuilds them first
        if t == 35:
            m = (np.arange(n) % 3 == 0).astype(np.uint8)
            g.new_game(m); o.new_game(m)
        if t == 50:
            a = synthetic_actions(game, n, t, seed=9)
            for x, y in zip(g.step(a, auto_reset=True), o.step(a, auto_reset=True)):
                assert np.array_equal(x, y)
        if t == 60:
            g.set_option(_abi.OPT_PIPELINE, 3)               # (no effect while a record ring is in force, and none on this call)
        if t in (65, 66):                                    # the unfused pair in between
            g.step_synthetic(77, t, auto_reset=True, stream=s.ptr)
            g.render_device(0, channels, stream=s.ptr)
            o.step(synthetic_actions(game, n, t, seed=77), auto_reset=True)
        g.render_step_synthetic(1337, t, channels=channels, auto_reset=True, stream=s.ptr)
        g.gather(stream=s.ptr)
        f, _ = g.device_buffer(_abi.BUF_FRAME)
        for k, i in enumerate(sample):
            hip.memcpy_dtod_async(hold + fb * (len(sample) * t + k), f + fb * i, fb, s)
        want.append([o.render_env(i, channels) for i in sample])
        ro = o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
        if t % 11 == 0:
            s.synchronize()
            p, _ = g.device_buffer(_abi.BUF_PACKED)
            hip.memcpy_dtoh(packed, p, 8 * n)
            assert np.array_equal(packed, pack_records(ro[0], ro[1], ro[2])), t
    s.synchronize()
    one = np.empty((H, W, channels), np.uint8)
    for t in range(T):
        for k, i in enumerate(sample):
            hip.memcpy_dtoh(one, hold + fb * (len(sample) * t + k), fb)
            assert np.array_equal(one, want[t][k]), (game, t, i)
    hip.free(hold)
    g.sync()
    _same_states(g, o, sample + list(range(0, n, max(1, n // 50))), "end")
    for x, y in zip(g.scalars(), o.scalars()):
        assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("n,channels,K", [(8192, 3, 0), (8192, 3, 1), (8192, 3, 4), (4096, 4, 1), (700, 3, 2), (20000, 3, 4)])
def test_fused_overlap_outputs_frames_and_gather(n, channels, K, hip_lib, oracle_lib):
    """Overlapped fused launches (TBX_OPT_FUSED_OVERLAP = 1) under the contract the header states: EVERY call's outputs (reward,
    done, lives, score, packed) and sampled frames are read by copies queued on the caller's stream right behind the call --
    i.e. before the next call, which may start while this call's rasteriser blocks still run -- and must equal the oracle's;
    TBX_BUF_FRAME / TBX_BUF_REWARD alternate between two addresses; with a gather (K = 1: a collective per step, now beside
    the next launch; K > 1: the record ring) the gathered block equals the oracle's records of the last K steps; a call with
    out_dev given, a stream-order call (option 2) and a host step in between join and re-enter.  8 192 envs + K = 4 is the
    per-GPU share of the strong-scaled headline batch."""
    from toybox_amd import hip
    from toybox_amd.parallel import pack_records
    game = "breakout"
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=21)
    H, W = g.height, g.width
    fb = H * W * channels
    g.set_option(_abi.OPT_FUSED_OVERLAP, _abi.FUSED_OVERLAP_ON)
    if K:
        g.set_option(_abi.OPT_GATHER_EVERY, K)
        g.gather_init(1, 0, g.gather_unique_id())
    assert g.get_option(_abi.OPT_FUSED_OVERLAP_ACTIVE) == 1
    sample = sorted({0, 1, 255, 256, n // 2, n - 1})
    s = hip.Stream()
    T = 64
    per = 4 * n * 3 + n + 8 * n                       # reward, lives, score (i32), done (u8), packed (u64)
    hold_o = hip.malloc(per * T)
    hold_f = hip.malloc(fb * len(sample) * T)
    side = hip.malloc(fb * n)                         # a caller-owned frame buffer for the out_dev call
    want_f, want_o, frame_addr, reward_addr = [], [], [], []
    gathered_checked = 0
    for t in range(T):
        if t == 23:
            a = synthetic_actions(game, n, t, seed=3)
            for x, y in zip(g.step(a, auto_reset=True), o.step(a, auto_reset=True)):
                assert np.array_equal(x, y)
        if t == 40:
            g.set_option(_abi.OPT_FUSED_OVERLAP, _abi.FUSED_OVERLAP_OFF)
        if t == 44:
            g.set_option(_abi.OPT_FUSED_OVERLAP, _abi.FUSED_OVERLAP_ON)
        if t == 31:                                   # the caller's own buffer: stream order for this call
            g.render_step_synthetic(1337, t, out_ptr=side, channels=channels, auto_reset=True, stream=s.ptr)
            f = side
        else:
            g.render_step_synthetic(1337, t, channels=channels, auto_reset=True, stream=s.ptr)
            f, _ = g.device_buffer(_abi.BUF_FRAME)
        if K:
            g.gather(stream=s.ptr)
        frame_addr.append(f)
        off = per * t
        for which, nb in ((_abi.BUF_REWARD, 4 * n), (_abi.BUF_LIVES, 4 * n), (_abi.BUF_SCORE, 4 * n), (_abi.BUF_DONE, n), (_abi.BUF_PACKED, 8 * n)):
            p, _ = g.device_buffer(which)
            if which == _abi.BUF_REWARD:
                reward_addr.append(p)
            hip.memcpy_dtod_async(hold_o + off, p, nb, s)
            off += nb
        for k, i in enumerate(sample):
            hip.memcpy_dtod_async(hold_f + fb * (len(sample) * t + k), f + fb * i, fb, s)
        want_f.append([o.render_env(i, channels) for i in sample])
        ro = o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
        want_o.append((ro[0], ro[2], ro[3], ro[1], pack_records(ro[0], ro[1], ro[2])))
        if K and (t + 1) % K == 0 and g.gather_fill() == 0:
            got = g.gather_host().reshape(K, -1)[:, :n]
            for j in range(K):
                assert np.array_equal(got[j], want_o[t - K + 1 + j][4]), (t, j)
            gathered_checked += 1
    s.synchronize()
    g.sync()                                          # (reports a ticket time-out of the wait kernel, if any)
    # the overlapped calls alternate between two frame buffers and two output sets
    run = list(range(2, 22))
    assert all(frame_addr[t] == frame_addr[t - 2] and frame_addr[t] != frame_addr[t - 1] for t in run[2:])
    assert all(reward_addr[t] == reward_addr[t - 2] and reward_addr[t] != reward_addr[t - 1] for t in run[2:])
    assert not K or gathered_checked >= T // K - 3
    buf = np.empty(per, np.uint8)
    one = np.empty((H, W, channels), np.uint8)
    for t in range(T):
        hip.memcpy_dtoh(buf, hold_o + per * t, per)
        rew, liv, sco = (buf[4 * n * k:4 * n * (k + 1)].view(np.int32) for k in range(3))
        don = buf[12 * n:13 * n]
        pk = buf[13 * n:].view(np.uint64)
        w = want_o[t]
        assert np.array_equal(rew, w[0]) and np.array_equal(liv, w[1]) and np.array_equal(sco, w[2]), t
        assert np.array_equal(don, w[3].astype(np.uint8)) and np.array_equal(pk, w[4]), t
        for k, i in enumerate(sample):
            hip.memcpy_dtoh(one, hold_f + fb * (len(sample) * t + k), fb)
            assert np.array_equal(one, want_f[t][k]), (t, i)
    for p in (hold_o, hold_f, side):
        hip.free(p)
    _same_states(g, o, sample + list(range(0, n, max(1, n // 40))), "end")
    g.close(); o.close()


@pytest.mark.parametrize("game", GAMES)
def test_rollout_synthetic_call_contract(game, lib):
    """tbx_rollout_synthetic(k) == k x (tbx_render_device ; tbx_step_synthetic) on either library: the chunk's frames (frame j = the
    state BEFORE step t0 + j), its step records, the last step's outputs, TBX_BUF_FRAME = the last frame, the state -- with calls of
    other kinds between chunks (a host step, a state write, new games, a single fused call, the option switched off and on)."""
    from support import read_buffer
    n, k = 9, 3
    a, b = Engine(game, n, lib=lib), Engine(game, n, lib=lib)
    for e in (a, b):
        e.seed(11)
        e.new_game()
    a.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_ON)
    H, W = a.height, a.width
    t = 0
    for c in range(9):
        if c == 3:
            act = synthetic_actions(game, n, 500, seed=3)
            for x, y in zip(a.step(act, auto_reset=True), b.step(act, auto_reset=True)):
                assert np.array_equal(x, y)
        if c == 4:
            for e in (a, b):
                st = e.get_state(2)
                e.set_state(5, st)
                e.new_game((np.arange(n) % 4 == 0).astype(np.uint8))
        if c == 5:
            a.render_step_synthetic(1337, t, channels=3, auto_reset=True)
            b.render_device(0, 3); b.step_synthetic(1337, t, auto_reset=True)
            t += 1
        if c == 2:
            a.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_SPAN)            # one rasteriser launch for the chunk's k x n frames
        if c == 4:
            a.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_PER_FRAME)
        if c == 6:
            a.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_OFF)
        if c == 7:
            a.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_AUTO)
        a.rollout_synthetic(1337, t, k, channels=3, auto_reset=True)
        frames = read_buffer(a, _abi.BUF_ROLLOUT_FRAMES, (k, n, H, W, 3))
        packed = read_buffer(a, _abi.BUF_ROLLOUT_PACKED, (k, n), np.uint64)
        for j in range(k):
            assert np.array_equal(frames[j], b.render(3)), (game, c, j)
            b.step_synthetic(1337, t + j, auto_reset=True)
            assert np.array_equal(packed[j], read_buffer(b, _abi.BUF_PACKED, (n,), np.uint64)), (game, c, j)
        for which, dt in ((_abi.BUF_REWARD, np.int32), (_abi.BUF_LIVES, np.int32), (_abi.BUF_SCORE, np.int32), (_abi.BUF_DONE, np.uint8), (_abi.BUF_PACKED, np.uint64)):
            assert np.array_equal(read_buffer(a, which, (n,), dt), read_buffer(b, which, (n,), dt)), (game, c, which)
        t += k
    a.sync(); b.sync()
    for i in range(n):
        assert bytes(a.get_state(i)) == bytes(b.get_state(i))
    with pytest.raises(ToyboxAmdError):
        a.rollout_synthetic(1337, 0, 0)
    with pytest.raises(ToyboxAmdError):
        a.rollout_synthetic(1337, 0, 2, channels=2)
    # under a K-step record ring the chunk has to be as long as the ring, and the ring empty
    a.set_option(_abi.OPT_GATHER_EVERY, 4)
    a.gather_init(1, 0, a.gather_unique_id())
    with pytest.raises(ToyboxAmdError):
        a.rollout_synthetic(1337, t, 3)
    a.step_synthetic(1337, t, auto_reset=True); a.gather()
    with pytest.raises(ToyboxAmdError):
        a.rollout_synthetic(1337, t + 1, 4)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game,n,channels,K,ring,form", [
    ("breakout", 8192, 3, 4, True, 1), ("breakout", 4096, 4, 4, False, 1), ("breakout", 700, 3, 3, True, 1), ("breakout", 20000, 3, 5, True, 1),
    ("breakout", 3000, 3, 2, False, 1), ("breakout", 8192, 3, 4, True, 3), ("breakout", 3000, 4, 3, False, 4), ("breakout", 40000, 3, 3, True, 4),
    ("space_invaders", 4096, 3, 4, True, 1), ("space_invaders", 900, 1, 3, False, 1), ("space_invaders", 9000, 4, 2, True, 1),
    ("space_invaders", 5000, 3, 4, True, 4)])
def test_rollout_chunks_equal_oracle(game, n, channels, K, ring, form, hip_lib, oracle_lib):
    """Rollout chunks on the device (one step launch on the step lane + the chunk's rasteriser launches, TBX_OPT_ROLLOUT_CHUNKS: form 1 =
    the engine's choice of rasteriser form, 3 = a launch per frame on two lanes, 4 = one launch for the chunk's k x n frames -- at most
    65 536 per launch, so 20 000 x 5 and 40 000 x 3 take several -- on one lane; chunk 12 of every case runs in the OTHER form)
    against the oracle's k single calls (Breakout: one multi-frame step launch per chunk; SpaceInvaders: the record of the current state
    and the k single-frame step launches back to back on the step lane): EVERY chunk's step records (all envs), last-step outputs and sampled frames, read by
    copies queued on the caller's stream right behind tbx_device_buffer (the lazy join) and therefore before the next chunk is
    issued; the two chunk buffers alternate; under a K-step ring the gathered block of every chunk; joins (a host step, a state
    write, a single fused call, the option off for a chunk) in between; final states.  8 192 envs + K = 4 ring is the per-GPU
    share of the strong-scaled headline batch."""
    from toybox_amd import hip
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=33)
    H, W = g.height, g.width
    fb = H * W * channels
    g.set_option(_abi.OPT_ROLLOUT_CHUNKS, form)
    if ring:
        for e in (g, o):
            e.set_option(_abi.OPT_GATHER_EVERY, K)
            e.gather_init(1, 0, e.gather_unique_id())
    assert g.get_option(_abi.OPT_ROLLOUT_CHUNKS_ACTIVE) == 1
    sample = sorted({0, 1, 255, 256, n // 2, n - 1})
    s = hip.Stream()
    chunks = 14
    per_out = 4 * n * 3 + n
    hold_p = hip.malloc(8 * n * K * chunks)
    hold_o = hip.malloc(per_out * chunks)
    hold_f = hip.malloc(fb * len(sample) * K * chunks)
    want_f, want_p, want_o, addr = [], [], [], []
    t = 0
    for c in range(chunks):
        if c == 4:
            a = synthetic_actions(game, n, 900, seed=3)
            for x, y in zip(g.step(a, auto_reset=True), o.step(a, auto_reset=True)):
                assert np.array_equal(x, y)
        if c == 6:
            st = o.get_state(7)
            g.set_state(1, st); o.set_state(1, st)
        if c == 8 and not ring:
            g.render_step_synthetic(1337, t, channels=channels, auto_reset=True, stream=s.ptr)
            o.render_step_synthetic(1337, t, channels=channels, auto_reset=True)
            t += 1
        other = _abi.ROLLOUT_CHUNKS_PER_FRAME if form == _abi.ROLLOUT_CHUNKS_SPAN else _abi.ROLLOUT_CHUNKS_SPAN
        g.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_OFF if c == 10 else other if c == 12 else form)
        g.rollout_synthetic(1337, t, K, channels=channels, auto_reset=True, stream=s.ptr)
        f, nb = g.device_buffer(_abi.BUF_ROLLOUT_FRAMES)
        assert nb == K * n * fb
        addr.append(f)
        pk, pb = g.device_buffer(_abi.BUF_ROLLOUT_PACKED)
        assert pb == 8 * K * n                                  # (one rank, records_per_rank = N: the ring's rows are N wide)
        hip.memcpy_dtod_async(hold_p + 8 * n * K * c, pk, 8 * n * K, s)
        off = per_out * c
        for which, nbytes in ((_abi.BUF_REWARD, 4 * n), (_abi.BUF_LIVES, 4 * n), (_abi.BUF_SCORE, 4 * n), (_abi.BUF_DONE, n)):
            p, _ = g.device_buffer(which)
            hip.memcpy_dtod_async(hold_o + off, p, nbytes, s)
            off += nbytes
        for j in range(K):
            for kk, i in enumerate(sample):
                hip.memcpy_dtod_async(hold_f + fb * ((c * K + j) * len(sample) + kk), f + fb * (j * n + i), fb, s)
        # the oracle: the k single calls
        fr, pr = [], []
        for j in range(K):
            fr.append([o.render_env(i, channels) for i in sample])
            o.step(synthetic_actions(game, n, t + j, seed=1337), auto_reset=True)
            q, _ = o.device_buffer(_abi.BUF_PACKED)
            pr.append(np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint64)), (n,)).copy())
            if ring:
                o.gather()
        want_f.append(fr); want_p.append(np.stack(pr))
        want_o.append([np.ctypeslib.as_array(C.cast(o.device_buffer(w)[0], C.POINTER(ct)), (n,)).copy()
                       for w, ct in ((_abi.BUF_REWARD, C.c_int32), (_abi.BUF_LIVES, C.c_int32), (_abi.BUF_SCORE, C.c_int32), (_abi.BUF_DONE, C.c_uint8))])
        if ring:
            assert g.gather_fill() == 0
            assert np.array_equal(g.gather_host().reshape(K, -1)[:, :n], o.gather_host().reshape(K, -1)[:, :n]), c
        t += K
    s.synchronize()
    g.sync()
    assert all(addr[c] != addr[c - 1] for c in (1, 2, 3, 12, 13)) and addr[2] == addr[0] and addr[3] == addr[1]
    pk = np.empty((K, n), np.uint64)
    buf = np.empty(per_out, np.uint8)
    one = np.empty((H, W, channels), np.uint8)
    for c in range(chunks):
        hip.memcpy_dtoh(pk, hold_p + 8 * n * K * c, 8 * n * K)
        assert np.array_equal(pk, want_p[c]), c
        hip.memcpy_dtoh(buf, hold_o + per_out * c, per_out)
        rew, liv, sco = (buf[4 * n * x:4 * n * (x + 1)].view(np.int32) for x in range(3))
        w = want_o[c]
        assert np.array_equal(rew, w[0]) and np.array_equal(liv, w[1]) and np.array_equal(sco, w[2]) and np.array_equal(buf[12 * n:], w[3]), c
        for j in range(K):
            for kk, i in enumerate(sample):
                hip.memcpy_dtoh(one, hold_f + fb * ((c * K + j) * len(sample) + kk), fb)
                assert np.array_equal(one, want_f[c][j][kk]), (c, j, i)
    for p in (hold_p, hold_o, hold_f):
        hip.free(p)
    _same_states(g, o, sample + list(range(0, n, max(1, n // 40))), "end")
    g.close(); o.close()


@pytest.mark.gpu
def test_space_invaders_short_load_only_while_states_are_plain(hip_lib, oracle_lib):
    """The canonical step kernel (si_load_canonical) derives enemy ids, the config's points and the lasers' constants instead of
    loading them; a state that carries other values -- on the grid, so the record rasteriser stays -- must take the engine back
    to the full load, in the batch step and the pipelined step alike."""
    game, n = "space_invaders", 600
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=12)
    for t in range(120):
        a = synthetic_actions(game, n, t)
        g.step(a, auto_reset=True); o.step(a, auto_reset=True)
    st = o.get_state(5)
    for k in range(st.n_enemies):
        st.enemies[k].points = 1000 + k                     # not the config's row scores
        st.enemies[k].id = 200 - k
    if st.n_enemy_lasers:
        st.enemy_lasers[0].speed = 1
    st.has_ship_laser = 1
    st.ship_laser.x, st.ship_laser.y, st.ship_laser.w, st.ship_laser.h = 100, 150, 3, 5
    st.ship_laser.t, st.ship_laser.movement, st.ship_laser.speed = 0, 0, 2
    for e in (g, o):
        e.set_state(5, st)
    assert g.get_option(_abi.OPT_RECORDS_ACTIVE) == 1
    g.set_option(_abi.OPT_PIPELINE, 3)                      # step_ahead picks its kernel by the same flag
    for t in range(120, 400):
        if t % 3 == 0:
            g.render_step_synthetic(1337, t, channels=3, auto_reset=False)
            o.render_step_synthetic(1337, t, channels=3, auto_reset=False)
        else:
            a = synthetic_actions(game, n, t)
            for x, y in zip(g.step(a), o.step(a)):
                assert np.array_equal(x, y), t
    g.sync()
    _same_states(g, o, range(0, n, 7), "end")
    assert bytes(g.get_state(5)) == bytes(o.get_state(5))
    assert np.array_equal(g.render_env(5, 3), o.render_env(5, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 3])
def test_a_caller_stream_may_be_destroyed_after_tbx_sync(mode, hip_lib, oracle_lib):
    """The library remembers the stream of the last call to order the next one behind it, so that stream has to live until
    the next call on the handle -- or until tbx_sync, which forgets it (include/toybox_amd.h, STREAM LIFETIME).  A caller that
    creates a stream for one pair of calls, calls tbx_sync and destroys the stream: the next call on the handle (another
    stream, or a host-pointer call) must neither fail nor lose its place in program order."""
    from toybox_amd import hip
    game, n = "breakout", 2048
    g, o = _pair(game, n, hip_lib, oracle_lib, seed=3)
    g.set_option(_abi.OPT_PIPELINE, mode)
    for t in range(12):
        s = hip.Stream()
        g.step_synthetic(1337, t, auto_reset=True, stream=s.ptr)
        g.render_device(0, 3, stream=s.ptr)
        g.sync()
        s.close()
        o.step(synthetic_actions(game, n, t, seed=1337), auto_reset=True)
        if t % 2:
            for x, y in zip(g.scalars(), o.scalars()):
                assert np.array_equal(x, y), t
        if t % 4 == 3:
            assert np.array_equal(g.render_env(t, 3), o.render_env(t, 3))
    _same_states(g, o, range(0, n, 97), "end")


@pytest.mark.gpu
def test_options_are_validated_and_reported(hip_lib, oracle_lib):
    for lib in (hip_lib, oracle_lib):
        e = Engine("amidar", 8, lib=lib)
        assert [e.get_option(k) for k in range(5)] == [0, 0, 0, 0, 1]
        e.set_option(_abi.OPT_RENDER_SPLIT, 7)
        assert e.get_option(_abi.OPT_RENDER_SPLIT) == 7
        for opt, val in ((_abi.OPT_PIPELINE, 4), (_abi.OPT_STEP_FORM, 3), (_abi.OPT_RENDER_SPLIT, -1), (99, 0), (_abi.OPT_AGENT_GENERIC, 2),
                         (_abi.OPT_ROLLOUT_CHUNKS, 5), (_abi.OPT_ROLLOUT_CHUNKS, -1), (_abi.OPT_FUSED_OVERLAP, 3)):
            with pytest.raises(ToyboxAmdError):
                e.set_option(opt, val)
        for val in (_abi.ROLLOUT_CHUNKS_PER_FRAME, _abi.ROLLOUT_CHUNKS_SPAN, _abi.ROLLOUT_CHUNKS_AUTO):     # (the rasteriser forms by name)
            e.set_option(_abi.OPT_ROLLOUT_CHUNKS, val)
            assert e.get_option(_abi.OPT_ROLLOUT_CHUNKS) == val
        e.close()


@pytest.mark.gpu
def test_engines_choice_of_pipelined_mode(hip_lib):
    """TBX_OPT_PIPELINE = 1: overlapped launches (3) for small Breakout / SpaceInvaders batches while no per-step gather is
    initialised, stream order (0) otherwise -- what include/toybox_amd.h says, read back through TBX_OPT_PIPELINE_ACTIVE."""
    want = {("breakout", 1024): 0, ("breakout", 4096): 3, ("breakout", 16384): 0, ("space_invaders", 1024): 3,
            ("space_invaders", 16384): 0, ("amidar", 4096): 0, ("gridworld", 4096): 0}
    for (game, n), mode in want.items():
        e = Engine(game, n, lib=hip_lib)
        assert e.get_option(_abi.OPT_PIPELINE_ACTIVE) == 0                     # the option is off by default
        e.set_option(_abi.OPT_PIPELINE, 1)
        assert e.get_option(_abi.OPT_PIPELINE_ACTIVE) == mode, (game, n)
        if mode:
            try:
                e.gather_init(1, 0, e.gather_unique_id())
            except ToyboxAmdError:
                e.close()
                continue                                                         # no librccl on this box
            assert e.get_option(_abi.OPT_PIPELINE_ACTIVE) == 0, (game, n)       # a gather: stream order
        e.close()


@pytest.mark.gpu
def test_engines_choice_of_rollout_chunks(hip_lib):
    """TBX_OPT_ROLLOUT_CHUNKS = 0: what the engines choose by game, batch size and gather (breakout.hip / space_invaders.hip,
    rollout_auto), read back through TBX_OPT_ROLLOUT_CHUNKS_ACTIVE: Breakout chunks up to 32 768 envs, under a K-step
    record ring only from 2 048; SpaceInvaders up to 8 192; never with a collective per step, never for Amidar / GridWorld; 1 / 3 / 4
    switch them on wherever the engine can, 2 off."""
    want = {("breakout", 1024): (1, 0), ("breakout", 2048): (1, 1), ("breakout", 8192): (1, 1), ("breakout", 32768): (1, 1), ("breakout", 40000): (0, 0),
            ("space_invaders", 4096): (1, 1), ("space_invaders", 8192): (1, 1), ("space_invaders", 12000): (0, 0), ("amidar", 4096): (0, 0),
            ("gridworld", 4096): (0, 0)}
    for (game, n), (plain, ring) in want.items():
        e = Engine(game, n, lib=hip_lib)
        assert e.get_option(_abi.OPT_ROLLOUT_CHUNKS_ACTIVE) == plain, (game, n)
        can = game in ("breakout", "space_invaders")
        for v in (_abi.ROLLOUT_CHUNKS_ON, _abi.ROLLOUT_CHUNKS_PER_FRAME, _abi.ROLLOUT_CHUNKS_SPAN):
            e.set_option(_abi.OPT_ROLLOUT_CHUNKS, v)
            assert e.get_option(_abi.OPT_ROLLOUT_CHUNKS_ACTIVE) == (1 if can else 0), (game, n, v)
        e.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_OFF)
        assert e.get_option(_abi.OPT_ROLLOUT_CHUNKS_ACTIVE) == 0
        e.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_AUTO)
        e.close()
        for every, expect in ((4, ring), (1, 0)):                                # a K-step ring / a collective per step
            e = Engine(game, n, lib=hip_lib)
            e.set_option(_abi.OPT_GATHER_EVERY, every)
            try:
                e.gather_init(1, 0, e.gather_unique_id())
            except ToyboxAmdError:
                e.close()
                break                                                            # no librccl on this box
            assert e.get_option(_abi.OPT_ROLLOUT_CHUNKS_ACTIVE) == expect, (game, n, every)
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game", GAMES)
def test_step1_frame_resident_kernel_paints(game, hip_lib, oracle_lib):
    """tbx_step1_frame = ToyboxBaseEnv.step in one call (envs/atari/base.py:126,109): on a one-env engine the resident kernel
    steps AND rasterises into pinned host memory.  Outputs and frames in all three formats against the oracle through game
    overs (auto-reset), interleaved with calls that stop the resident kernel (state reads, a batch render, a new game); then
    the same entry point on a batch engine (launch + copy path)."""
    g, o = Engine(game, 1, lib=hip_lib), Engine(game, 1, lib=oracle_lib)
    for e in (g, o):
        e.seed(21); e.new_game()
    legal = LEGAL[game]
    rng = np.random.default_rng(4)
    dones = 0
    for t in range(1500):
        a = int(legal[int(rng.integers(len(legal)))]) if game != "breakout" or t % 9 else 1
        ch = (1, 3, 4)[t % 3]
        x, y = g.step1_frame(0, a, ch, auto_reset=True), o.step1_frame(0, a, ch, auto_reset=True)
        assert x[:4] == y[:4], (t, x[:4], y[:4])
        assert x[4].shape == (g.height, g.width, ch) and np.array_equal(x[4], y[4]), (t, ch)
        dones += int(x[1])
        if t % 211 == 210:
            assert bytes(g.get_state(0)) == bytes(o.get_state(0)), t            # stops the resident kernel; it restarts on demand
            assert np.array_equal(g.render(3), o.render(3))
        if t == 700:
            g.new_game(); o.new_game()
    if game == "breakout":
        assert dones > 0
    with pytest.raises(ToyboxAmdError):
        g.step1_frame(0, 0, 2)
    # a custom-bricks / off-grid state keeps working (Breakout: wave-per-env step + record built on the spot; SpaceInvaders:
    # state-reading painter)
    if game in ("breakout", "space_invaders"):
        st = o.get_state(0)
        if game == "breakout":
            st.bricks[3].w += 2.0
        else:
            st.enemies[2].x += 3
        for e in (g, o):
            e.set_state(0, st)
        for t in range(60):
            a = int(legal[t % len(legal)])
            x, y = g.step1_frame(0, a, 3), o.step1_frame(0, a, 3)
            assert x[:4] == y[:4] and np.array_equal(x[4], y[4]), t
    # batch engine: the same call is tbx_apply_input + tbx_render_env
    gb, ob = Engine(game, 5, lib=hip_lib), Engine(game, 5, lib=oracle_lib)
    for e in (gb, ob):
        e.seed(2); e.new_game()
    for t in range(40):
        env, a = t % 5, int(legal[t % len(legal)])
        x, y = gb.step1_frame(env, a, 3), ob.step1_frame(env, a, 3)
        assert x[:4] == y[:4] and np.array_equal(x[4], y[4]), t
