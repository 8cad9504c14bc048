"""Multi-rank path on CPU, world size 2, over the oracle's ABI; both must equal a single-process run over the whole batch
(results independent of the number of ranks):
  * the product path: communicator id made by rank 0, handed over through toybox_amd.parallel.exchange_unique_id, collective
    tbx_gather_init, one tbx_gather per step, tbx_gather_reduce_max (the oracle restates these calls over shared memory, the
    HIP library over RCCL);
  * the host fallback: the same records through a gloo process group."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from toybox_amd.parallel import pack_records, shard_range, unpack_records

WORKER = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from toybox_amd import Engine, _abi
from toybox_amd.parallel import ShardedBatch
from support import synthetic_actions
lib = ctypes.CDLL(os.path.join({root!r}, "oracle", "liboracle.so")); _abi.bind(lib)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
sb = ShardedBatch(lambda n: Engine("breakout", n, lib=lib), {n}, rank=dist.get_rank(), world=2, host_dist=dist)
tot = np.zeros({n}, np.int64); dn = np.zeros({n}, np.int64)
for t in range({steps}):
    r, d, l = sb.step_host(synthetic_actions("breakout", {n}, t))
    tot += r; dn += d
if dist.get_rank() == 0:
    np.save({out!r}, np.stack([tot, dn, l.astype(np.int64)]))
dist.barrier(); dist.destroy_process_group()
"""


def test_shard_range_partition():
    for n, w in ((10, 3), (65536, 8), (7, 8), (262144, 8)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_record_packing_roundtrip():
    r = np.array([0, 7, 2**31 - 1, 4], np.int32)
    d = np.array([0, 1, 0, 1], bool)
    l = np.array([5, 0, 255, 3])
    rr, dd, ll = unpack_records(pack_records(r, d, l))
    assert np.array_equal(rr, r) and np.array_equal(dd, d) and np.array_equal(ll, l)


def test_two_rank_gloo_equals_single_process(oracle_lib, tmp_path):
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from support import synthetic_actions
    from toybox_amd import Engine
    from toybox_amd.parallel import ShardedBatch
    n, steps, port = 37, 700, 29000 + os.getpid() % 2000
    out = str(tmp_path / "r0.npy")
    src = WORKER.format(root=ROOT, port=port, n=n, steps=steps, out=out)
    script = tmp_path / "worker.py"
    script.write_text(src)
    procs = [subprocess.Popen([sys.executable, str(script), str(r)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    sb = ShardedBatch(lambda k: Engine("breakout", k, lib=oracle_lib), n)
    tot = np.zeros(n, np.int64)
    dn = np.zeros(n, np.int64)
    for t in range(steps):
        r, d, l = sb.step_host(synthetic_actions("breakout", n, t))
        tot += r
        dn += d
    assert np.array_equal(got[0], tot) and np.array_equal(got[1], dn) and np.array_equal(got[2], l.astype(np.int64))
    assert tot.sum() > 0


ABI_WORKER = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from toybox_amd import Engine, _abi
from toybox_amd.parallel import ShardedBatch, world_from_env
from support import synthetic_actions
lib = ctypes.CDLL(os.path.join({root!r}, "oracle", "liboracle.so")); _abi.bind(lib)
rank, world, _ = world_from_env()
sb = ShardedBatch(lambda n: Engine({game!r}, n, lib=lib), {n}, rank=rank, world=world)
tot = np.zeros({n}, np.int64); dn = np.zeros({n}, np.int64)
for t in range({steps}):
    if t % 2:
        r, d, l = sb.step_host(synthetic_actions({game!r}, {n}, t))
    else:                                   # device-resident form: in-kernel actions by global index + asynchronous gather
        sb.step_synthetic(1337, t)
        r, d, l = sb.gathered()
    tot += r; dn += d
slowest = sb.max_over_ranks(10.0 + rank)
assert slowest == 10.0 + world - 1, slowest
if rank == 0:
    np.save({out!r}, np.stack([tot, dn, l.astype(np.int64)]))
sb.close()
"""


import pytest  # noqa: E402


@pytest.mark.parametrize("game,world", [("breakout", 2), ("amidar", 3)])
def test_multi_rank_gather_through_the_abi_equals_single_process(game, world, oracle_lib, tmp_path):
    """n = 37 does not divide: shards of unequal size, padded slots in the gathered layout."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from support import synthetic_actions
    from toybox_amd import Engine
    from toybox_amd.parallel import ShardedBatch
    n, steps = 37, 400
    out = str(tmp_path / "r0.npy")
    script = tmp_path / "worker.py"
    script.write_text(ABI_WORKER.format(root=ROOT, n=n, steps=steps, out=out, game=game))
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="1", TBX_RDZV_DIR=str(tmp_path))
    env.pop("TBX_RDZV_KEY", None)           # the default key (port + run id + parent pid) must do
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    sb = ShardedBatch(lambda k: Engine(game, k, lib=oracle_lib), n)
    tot = np.zeros(n, np.int64)
    dn = np.zeros(n, np.int64)
    for t in range(steps):
        r, d, l = sb.step_host(synthetic_actions(game, n, t, seed=1337 if t % 2 == 0 else 1337))
        tot += r
        dn += d
    assert np.array_equal(got[0], tot) and np.array_equal(got[1], dn) and np.array_equal(got[2], l.astype(np.int64))
    assert not os.listdir(str(tmp_path)) or all(not f.startswith("tbx_rccl_id_") for f in os.listdir(str(tmp_path))), "id file left behind"


RING_WORKER = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from toybox_amd import Engine, _abi
from toybox_amd.parallel import ShardedBatch, world_from_env
lib = ctypes.CDLL(os.path.join({root!r}, "oracle", "liboracle.so")); _abi.bind(lib)
rank, world, _ = world_from_env()
sb = ShardedBatch(lambda n: Engine({game!r}, n, lib=lib), {n}, rank=rank, world=world, gather_every={K})
assert sb.engine.gather_every() == {K}
tot = np.zeros({n}, np.int64); dn = np.zeros({n}, np.int64); calls = 0
for t in range({steps}):
    sb.step_synthetic(1337, t)
    if (t + 1) % {K} == 0:                  # the collective went out with this step: K steps' records, oldest first
        for r, d, l in sb.gathered():
            tot += r; dn += d
        calls += 1
assert calls == {steps} // {K}
if rank == 0:
    np.save({out!r}, np.stack([tot, dn, l.astype(np.int64)]))
sb.close()
"""


@pytest.mark.parametrize("game,world,K", [("breakout", 2, 4), ("space_invaders", 3, 5)])
def test_multi_rank_ring_gather_equals_single_process(game, world, K, oracle_lib, tmp_path):
    """The K-step record ring (TBX_OPT_GATHER_EVERY) over world > 1: every rank's K x width ring travels with ONE collective
    per K steps; summed over the run the records equal a single-process run over the whole batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from support import synthetic_actions
    from toybox_amd import Engine
    from toybox_amd.parallel import ShardedBatch
    n, steps = 37, 40 * K
    out = str(tmp_path / "r0.npy")
    script = tmp_path / "worker.py"
    script.write_text(RING_WORKER.format(root=ROOT, n=n, steps=steps, out=out, game=game, K=K))
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="2", TBX_RDZV_DIR=str(tmp_path))
    env.pop("TBX_RDZV_KEY", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    sb = ShardedBatch(lambda k: Engine(game, k, lib=oracle_lib), n)
    tot = np.zeros(n, np.int64)
    dn = np.zeros(n, np.int64)
    for t in range(steps):
        r, d, l = sb.step_host(synthetic_actions(game, n, t, seed=1337))
        tot += r
        dn += d
    assert np.array_equal(got[0], tot) and np.array_equal(got[1], dn) and np.array_equal(got[2], l.astype(np.int64))


def test_exchange_unique_id_times_out(tmp_path):
    from toybox_amd.parallel import exchange_unique_id
    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 2, lambda: b"", key="nobody", timeout=0.2, directory=str(tmp_path))


def test_mixed_batch_matches_separate_engines(oracle_lib):
    """BASELINE config 5 shape: a mixed Breakout + Amidar + SpaceInvaders batch is three homogeneous segments whose
    seeds / synthetic actions use the global env index."""
    from support import synthetic_actions
    from toybox_amd import Engine
    from toybox_amd.parallel import MixedBatch
    games, per = ["breakout", "amidar", "space_invaders"], 5
    mb = MixedBatch(games, per, engine_factory=lambda g, n: Engine(g, n, lib=oracle_lib))
    singles = []
    for i, g in enumerate(games):
        e = Engine(g, per, lib=oracle_lib)
        e.seed(1234 + i * per)
        e.new_game()
        singles.append(e)
    for t in range(300):
        mb.step_synthetic(1337, t)
        for i, (g, e) in enumerate(zip(games, singles)):
            e.step(synthetic_actions(g, per, t, seed=1337, env_offset=i * per), auto_reset=True)
    for me, se in zip(mb.engines, singles):
        for k in range(per):
            assert bytes(me.get_state(k)) == bytes(se.get_state(k))
    assert mb.n_envs == 15 and mb.frame_bytes(3) == per * 3 * (160 * 240 + 250 * 160 + 210 * 320)


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["breakout", "space_invaders", "amidar"])
def test_gpu_shard_decomposition_independence(game, hip_lib):
    """The 8-GPU layout of BASELINE's headline config on one GPU: eight engines of 8192 envs, each seeded and fed in-kernel
    actions by GLOBAL env index, end in exactly the states of one 65 536-env engine (no result depends on the rank count)."""
    from toybox_amd import Engine
    n, shards, steps = 65536, 8, 250
    whole = Engine(game, n, lib=hip_lib)
    whole.seed(1234)
    whole.new_game()
    per = n // shards
    parts = []
    for r in range(shards):
        e = Engine(game, per, lib=hip_lib)
        e.seed(1234 + r * per)
        e.new_game()
        parts.append(e)
    for t in range(steps):
        whole.step_synthetic(1337, t, env_offset=0, auto_reset=True)
        for r, e in enumerate(parts):
            e.step_synthetic(1337, t, env_offset=r * per, auto_reset=True)
    whole.sync()
    sw = whole.scalars()
    for r, e in enumerate(parts):
        e.sync()
        for x, y in zip(e.scalars(), sw):
            assert np.array_equal(x, y[r * per:(r + 1) * per]), (game, r)
        for k in (0, 1, per // 2, per - 1):
            assert bytes(e.get_state(k)) == bytes(whole.get_state(r * per + k)), (game, r, k)
    for e in parts + [whole]:
        e.close()


def _file_world_worker(rank, world, key, q):
    from toybox_amd.parallel import FileWorld
    fw = FileWorld(rank, world, key=key)
    out = []
    for i in range(40):
        fw.barrier()
        out.append(fw.allreduce_max(rank * 10 + i))
    q.put((rank, out))


def test_file_world_barrier_and_max(tmp_path, monkeypatch):
    """bench.py's fallback when no RCCL communicator can be made: ranks of one node meet through files."""
    import multiprocessing as mp
    monkeypatch.setenv("TBX_RDZV_DIR", str(tmp_path))
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_file_world_worker, args=(r, world, "t%d" % os.getpid(), q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    assert all(out == [10 * (world - 1) + i for i in range(40)] for _, out in res)
    assert len(list(tmp_path.iterdir())) <= world          # only the last round's files are left


def test_bench_launcher_dry_run_walks_the_n_process_path():
    """bench.py --gpus N starts its own ranks when no launcher did (subproc_vec_env.py:49-74 is the reference's N-worker
    launch): --dry-run takes that path end to end without a GPU -- N processes, the communicator id from rank 0 to every rank
    through the rendezvous file, max-over-ranks barriers around every region, teardown, ONE JSON line from rank 0 -- under both
    readings of the metric (weak: per-GPU batch; strong: the batch cut into contiguous shards that cover it exactly)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    for scaling, envs, world in (("strong", 65536, 8), ("weak", 1000, 3)):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run", "--steps", "4",
                            "--repeats", "2", "--scaling", scaling, "--envs", str(envs)], capture_output=True, text=True, timeout=300,
                           cwd="/tmp", env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, p.stdout
        out = json.loads(lines[0])
        assert out["dry_run"] and out["n_gpus"] == world and out["scaling"] == scaling
        total = envs if scaling == "strong" else envs * world
        assert out["config"]["envs_total"] == total == out["config"]["envs_covered"]
        assert out["id_exchange"].startswith("ok: %d ranks" % world)
        # the slowest rank sleeps 0.2 ms x world per step: the max over ranks is what is reported
        assert out["ms_per_step"] >= 0.2 * world


def test_bench_launcher_world_size_mismatch_is_an_error():
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True,
                       timeout=120, cwd="/tmp", env=env)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


BENCH_FLOW_WORKER = r"""
# One rank of `bench.py --gpus N` with the CPU checker behind bench.ENGINE_FACTORY and a stand-in for toybox_amd.hip behind
# bench.HIP_MODULE: everything main() does for N > 1 -- both readings, a communicator each, the verified exchange, max-over-ranks
# regions, rank 0's JSON line -- runs for real; only the kernels are the oracle's.
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
import bench
from toybox_amd import Engine, _abi
lib = ctypes.CDLL(os.path.join({root!r}, "oracle", "liboracle.so")); _abi.bind(lib)

class Ev:
    def __init__(self): self.t = 0.0
    def record(self, stream): self.t = time.perf_counter()
    def elapsed_ms(self, other): return max(1e-3, 1000.0 * (other.t - self.t))
    def close(self): pass
class St:
    ptr = 0
    def synchronize(self): pass
    def close(self): pass
class FakeHip:
    Event, Stream = Ev, St
    @staticmethod
    def synchronize(): pass
    @staticmethod
    def set_device(i): pass
    @staticmethod
    def memcpy_dtoh(dst, src, nbytes): ctypes.memmove(dst.ctypes.data, src, nbytes)
bench.ENGINE_FACTORY = lambda game, n, device: Engine(game, n, lib=lib)
bench.HIP_MODULE = FakeHip
sys.argv = ["bench.py"] + {argv!r}
sys.exit(bench.main())
"""


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["--scaling", "weak", "--gather-every", "1", "--loop", "pair"])])
def test_bench_n_process_flow_end_to_end_on_the_checker(world, extra, oracle_lib, tmp_path):
    """The first hardware run of `bench.py --gpus N > 1` is the driver's: everything in it that is not a kernel is walked here --
    value = the strong reading (envs in total) with the weak one beside it, a K-step ring communicator per reading, the verified
    exchange, share_of_linear as a fraction -- over the oracle's shared-memory gather."""
    import json
    # world 2 also carries the CPU arm: the north star wants it "in the same run" at every GPU count (rank 0, bounded sample)
    argv = ["--gpus", str(world), "--envs", "96", "--steps", "6", "--warmup", "2", "--repeats", "2", "--preroll", "300"] + \
           (["--cpu-seconds", "0.3"] if world == 2 else ["--no-cpu-baseline"]) + extra
    script = tmp_path / "worker.py"
    script.write_text(BENCH_FLOW_WORKER.format(root=ROOT, argv=argv))
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="3", TBX_RDZV_DIR=str(tmp_path))
    env.pop("TBX_RDZV_KEY", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True) for r in range(world)]
    out0 = procs[0].communicate(timeout=600)[0]
    for p in procs:
        assert p.wait(timeout=600) == 0, out0[-2000:]
    line = json.loads([ln for ln in out0.splitlines() if ln.startswith("{")][-1])
    strong_is_value = "--scaling" not in extra
    assert line["n_gpus"] == world and line["scaling"] == ("strong" if strong_is_value else "weak")
    assert line["rccl"]["nranks"] == world and line["rccl"]["verified"] is True
    assert line["rccl"]["gather_every"] == (4 if strong_is_value else 1)
    other = line["weak" if strong_is_value else "strong"]
    assert other["rccl"]["verified"] is True and other["rccl"]["nranks"] == world
    s_, w_ = (line, other) if strong_is_value else (other, line)
    assert s_["config"]["envs_total"] == 96 if strong_is_value else s_["envs_total"] == 96
    assert (w_["envs_total"] if strong_is_value else w_["config"]["envs_total"]) == 96 * world
    assert abs(line["share_of_linear"] - s_["value"] / w_["value"]) < 1e-9 and 0 < line["share_of_linear"]
    assert line["loop"]["form"] == "pair"                      # (the checker reports no fused launch)
    assert line["check"]["frames_played"] > 300
    assert line["metric_version"]["version"] == 2
    if world == 2:
        cb = line["cpu_baseline"]
        assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "96" in cb["sample"]
        assert line["cpu_config1"]["step_only"] > 0
    else:
        assert "cpu_baseline" not in line
    rf = line["roofline"]
    assert rf["avg_launch_ms"] > 0 and rf["launches_timed"] >= 1 and "never" in rf["timing"]


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_n_process_flow_under_torch_distributed_run(oracle_lib, tmp_path):
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- with the checker behind the engine: the ranks take RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the launcher (which also sets OMP_NUM_THREADS = 1: the CPU arm names its own thread count), the
    rendezvous key is derived from MASTER_ADDR:MASTER_PORT, rank 0 prints the one line."""
    import json
    pytest.importorskip("torch")
    world = 2
    argv = ["--gpus", str(world), "--envs", "96", "--steps", "6", "--warmup", "2", "--repeats", "2", "--preroll", "300", "--cpu-seconds", "0.3"]
    script = tmp_path / "worker.py"
    script.write_text(BENCH_FLOW_WORKER.format(root=ROOT, argv=argv))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TBX_RDZV_KEY", "MASTER_ADDR", "MASTER_PORT")}
    env["TBX_RDZV_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["rccl"]["nranks"] == world and line["rccl"]["verified"] is True
    assert line["weak"]["rccl"]["verified"] is True and 0 < line["share_of_linear"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1


@pytest.mark.parametrize("strict", [False, True])
def test_bench_falls_back_to_the_host_transport_when_rccl_fails(strict, oracle_lib, tmp_path):
    """N > 1 and the RCCL communicator cannot be made (forced here with TBX_BENCH_NO_RCCL): the ranks agree on that through the
    rendezvous directory and ALL fall back to the host transport -- the line says `rccl: null` and names the error -- so that
    the first hardware run yields a labelled line instead of none; --strict-rccl keeps the failed run (rc 4, no line)."""
    import json
    world = 2
    argv = ["--gpus", str(world), "--envs", "64", "--steps", "4", "--warmup", "1", "--repeats", "2", "--preroll", "300", "--settle", "2",
            "--no-cpu-baseline", "--no-extras"] + (["--strict-rccl"] if strict else [])
    script = tmp_path / "worker.py"
    script.write_text(BENCH_FLOW_WORKER.format(root=ROOT, argv=argv))
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="5", TBX_RDZV_DIR=str(tmp_path), TBX_BENCH_NO_RCCL="1")
    env.pop("TBX_RDZV_KEY", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True) for r in range(world)]
    out0, err0 = procs[0].communicate(timeout=600)
    codes = [p.wait(timeout=600) for p in procs]
    if strict:
        assert codes == [4] * world and not [ln for ln in out0.splitlines() if ln.startswith("{")]
        return
    assert codes == [0] * world, err0[-2000:]
    line = json.loads([ln for ln in out0.splitlines() if ln.startswith("{")][-1])
    assert line["rccl"] is None and line["gather"]["transport"] == "host" and line["gather"]["verified"] is True
    assert "TBX_BENCH_NO_RCCL" in line["gather"]["fallback_from_rccl"] and line["gather"]["nranks"] == world
    assert "falling back to the host transport" in err0


class _FakeEvent:
    log = []

    def __init__(self):
        self.at = None

    def record(self, stream):
        self.at = len(_FakeEvent.log)
        _FakeEvent.log.append(("event", id(self)))

    def elapsed_ms(self, other):
        # "time" = launches between the two events
        return float(sum(1 for x in _FakeEvent.log[self.at:other.at] if x[0] == "launch"))

    def close(self):
        pass


class _FakeEngine:
    def step_synthetic(self, *a, **k):
        _FakeEvent.log.append(("step",))

    def render_device(self, *a, **k):
        _FakeEvent.log.append(("launch",))

    def render_step_synthetic(self, *a, **k):
        _FakeEvent.log.append(("launch",))

    def gather(self, **k):
        pass

    def get_option(self, option):          # (stream order: neither overlapped fused launches nor rollout chunks)
        return 0


@pytest.mark.parametrize("fused,K", [(True, 20), (True, 5), (True, 1), (False, 20), (False, 3)])
def test_bench_launch_timing_brackets_runs_never_next_to_a_sync(fused, K):
    """roofline.avg_launch_ms of bench.py (bracketing single launches incl. the first one after a
    synchronisation made a kernel 'longer' than the step that contains it).  Fused loop: chained events around runs of up to 8
    back-to-back launches, none of the first two launches of a region; pair loop: single rasteriser launches, the same rule;
    the counters start again with every region."""
    import bench

    class Hip:
        Event = _FakeEvent

    class St:
        ptr = 0

    _FakeEvent.log = []
    R = 3
    loop = bench.Loop(_FakeEngine(), Hip, St(), 0, 3, False, True, K, R, fused=fused)
    for t in range(4):
        loop.full_step(t)                                   # warm-up: nothing is timed before arm()
    assert not any(x[0] == "event" for x in _FakeEvent.log)
    loop.arm()
    region_starts = []
    for r in range(R):
        region_starts.append(len(_FakeEvent.log))
        loop.begin_region()
        for i in range(K):
            loop.full_step(100 * r + i)
    per, covered = loop.launch_ms()
    run, skip = loop.run, loop.skip
    assert run == max(1, min(8, K - 2)) and skip == max(0, min(2, K - run))
    if fused:
        assert covered == R * ((K - skip) // run) * run and all(v == 1.0 for v in per)     # run launches between the events / run
    else:
        assert covered == R * ((K - skip + run - 1) // run) and all(v == 1.0 for v in per)
    # no event before the `skip`-th launch of a region
    for start in region_starts:
        seen = 0
        for x in _FakeEvent.log[start:]:
            if x[0] == "launch":
                seen += 1
            if x[0] == "event":
                assert seen >= skip
                break


def test_bench_chunk_form_runs_whole_chunks_and_times_one_span_per_region():
    """bench.py's fused loop as rollout chunks (tbx_rollout_synthetic, round 6): only when every phase is a whole number of chunks and
    the ring (if any) is as deep as the chunk; a region is K / k chunk calls; its launch timing is ONE span from behind the first
    chunk to behind the last one, each mark preceded by the address request that joins the caller's stream (lazy join)."""
    import bench
    from toybox_amd import _abi

    class Eng(_FakeEngine):
        ring = 4

        def get_option(self, option):
            return 1 if option == _abi.OPT_ROLLOUT_CHUNKS_ACTIVE else 0

        def gather_every(self):
            return self.ring

        def rollout_synthetic(self, seed, t0, k, **kw):
            _FakeEvent.log.append(("chunk", t0, k))
            for _ in range(k):
                _FakeEvent.log.append(("launch",))

        def device_buffer(self, which):
            _FakeEvent.log.append(("join", which))
            return 1, 1

    class Hip:
        Event = _FakeEvent

    class St:
        ptr = 0

    K, R, k = 20, 3, 4
    _FakeEvent.log = []
    loop = bench.Loop(Eng(), Hip, St(), 0, 3, True, True, K, R, fused=True)
    assert not loop.use_chunks(k, (18,)) and not loop.use_chunks(5, (K,))   # the timed phase / the ring do not fit
    assert not bench.Loop(Eng(), Hip, St(), 0, 3, True, True, K, R, fused=False).use_chunks(k, (44, K))            # the two-launch loop
    assert loop.use_chunks(k, (44, K)) and loop.chunk_k == k and loop.overlapped
    loop.many_steps(0, 44)                                  # settle + warm-up: nothing is timed
    assert [x for x in _FakeEvent.log if x[0] == "chunk"] == [("chunk", 4 * c, 4) for c in range(11)]
    assert not any(x[0] in ("event", "join") for x in _FakeEvent.log)
    loop.arm()
    for r in range(R):
        loop.begin_region()
        at = len(_FakeEvent.log)
        loop.many_steps(100 * r, K)
        kinds = [x[0] for x in _FakeEvent.log[at:]]
        assert kinds.count("chunk") == K // k and kinds.count("event") == 2 and kinds.count("join") == 2
        first_ev, last_ev = [i for i, x in enumerate(kinds) if x == "event"]
        assert kinds[first_ev - 1] == "join" and kinds[last_ev - 1] == "join" and last_ev == len(kinds) - 1
        assert kinds[:first_ev].count("launch") == k        # behind the first chunk
    per, covered = loop.launch_ms()
    assert covered == R * (K - k) and all(v == 1.0 for v in per)


@pytest.mark.gpu
@pytest.mark.parametrize("every", [1, 3])
def test_gpu_host_transport_gather_between_two_engines_on_one_device(every, hip_lib):
    """TBX_OPT_GATHER_TRANSPORT = 1 (SURVEY 8e's fallback, and the way two ranks can share ONE GPU): two engines = two ranks, one
    thread each (a host-transport collective blocks its caller until every rank has arrived), the same id, shards of unequal
    size; the gathered block of every collective must hold both ranks' records in rank order -- checked against one engine that
    owns the whole batch -- with one collective per step and with a 3-step ring; the max-reduction and what the engine reports
    about the transport."""
    import threading
    from toybox_amd import Engine, _abi
    from toybox_amd.parallel import shard_range, unpack_records
    game, n_total, world, steps = "breakout", 1000, 2, 12
    spans = [shard_range(n_total, world, r) for r in range(world)]
    width = max(e - s for s, e in spans)
    ref = Engine(game, n_total, lib=hip_lib)
    ref.seed(1234); ref.new_game()
    want = []
    for t in range(steps):
        ref.step_synthetic(1337, t, env_offset=0, auto_reset=True)
        ref.sync()
        p, _ = ref.device_buffer(_abi.BUF_PACKED)
        rec = np.empty(n_total, np.uint64)
        from toybox_amd import hip
        hip.memcpy_dtoh(rec, p, 8 * n_total)
        want.append(rec)
    ref.close()
    engines = []
    for r, (s, e_) in enumerate(spans):
        e = Engine(game, e_ - s, lib=hip_lib)
        e.seed(1234 + s); e.new_game()
        e.set_option(_abi.OPT_GATHER_TRANSPORT, _abi.GATHER_HOST)
        e.set_option(_abi.OPT_GATHER_EVERY, every)
        engines.append(e)
    uid = engines[0].gather_unique_id()
    got, errors, maxima = [[] for _ in spans], [], [None] * world

    def rank_main(r):
        try:
            e = engines[r]
            e.gather_init(world, r, uid, records_per_rank=width)
            assert e.gather_nranks() == world and e.gather_library().startswith("host:")
            for t in range(steps):
                e.step_synthetic(1337, t, env_offset=spans[r][0], auto_reset=True)
                e.gather()
                if (t + 1) % every == 0:
                    got[r].append(e.gather_host().copy())
            maxima[r] = e.gather_reduce_max(10.0 + r)
        except Exception as ex:                                # noqa: BLE001
            errors.append((r, repr(ex)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    assert maxima == [10.0 + world - 1] * world
    for r in range(world):
        assert len(got[r]) == steps // every
        for c, block in enumerate(got[r]):
            block = block.reshape(world, every, width)
            for j in range(every):
                t = c * every + j
                for q, (s, e_) in enumerate(spans):
                    assert np.array_equal(block[q, j, :e_ - s], want[t][s:e_]), (r, t, q)
                    assert not block[q, j, e_ - s:].any()
    lives = unpack_records(got[0][-1].reshape(world, every, width)[1, -1, :spans[1][1] - spans[1][0]])[2]
    assert lives.min() >= 1
    for e in engines:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["own", "torch.distributed.run"])
def test_gpu_bench_n_process_flow_on_one_device_with_the_host_transport(launcher, tmp_path):
    """The dress rehearsal of `bench.py --gpus N` that needs no second GPU: bench.py's own spawner starts two
    fresh rank processes (before anything touches the GPU), both on device 0, and the whole N > 1 flow runs on real HIP --
    engines for contiguous shards, both readings of the metric with a communicator each, the verified exchange, ring
    bookkeeping, max-over-ranks regions, rank 0's JSON line with the CPU arm -- everything except RCCL's wire, which is
    replaced by the host transport and labelled as such."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TBX_RDZV_KEY")}
    env["TBX_RDZV_DIR"] = str(tmp_path)
    # "torch.distributed.run": the driver's own launch line for N > 1 (the launcher exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
    if launcher != "own":
        pytest.importorskip("torch")
        env.pop("MASTER_ADDR", None); env.pop("MASTER_PORT", None)
    head = [sys.executable] if launcher == "own" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    p = subprocess.run(head + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gather", "host", "--one-device", "--envs", "4096",
                               "--steps", "6", "--warmup", "2", "--repeats", "2", "--preroll", "300", "--settle", "4", "--cpu-seconds", "0.5"],
                       capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["envs_total"] == 4096 and line["config"]["envs_per_gpu"] == 2048
    assert line["rccl"] is None
    g = line["gather"]
    assert g["transport"] == "host" and g["nranks"] == 2 and g["verified"] is True and g["gather_every"] == 4 and g["lib"].startswith("host:")
    assert line["weak"]["gather"]["transport"] == "host" and line["weak"]["envs_total"] == 8192 and line["weak"]["rccl"] is None
    assert 0 < line["share_of_linear"] < 1.5
    assert "HOST-STAGED" in line["config"]["parallelism"] and not line["config"]["parallelism"].startswith("HOST-STAGED FALLBACK")   # (asked for, not fallen back to)
    assert [r_["rank"] for r_ in line["ranks"]] == [0, 1] and line["ranks"][0]["pci"] == line["ranks"][1]["pci"]      # --one-device: both ranks on device 0
    assert line["loop"]["form"] == "fused" and line["roofline"]["avg_launch_ms"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["check"]["mean_score"] > 0


@pytest.mark.gpu
def test_gpu_bench_default_line_carries_every_arm():
    """The driver's own N = 1 command (`bench.py --gpus 1 --steps 20 --warmup 5`; the CPU arm cut to a second here): ONE JSON line
    whose roofline timing is consistent with the step it sits in, with BASELINE configs 2-5, the strong-scaling probe incl. the
    policy loop, the agent path of every game in both observation forms, and the CPU oracle beside it."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TBX_RDZV_KEY")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["unit"] == "env-steps/s" and j["value"] > 1e7
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and 0.5 < rf["frac"] < 1.0 and rf["launches_timed"] >= 40
    assert rf["avg_launch_ms"] <= 1.01 * j["ms_per_step"]            # a kernel cannot take longer than the step that contains it
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert j["serialised"]["value"] > 0 and j["scaling_strong"]["policy_loop"]["share_of_linear"] > 0.5
    cfg = j["configs"]
    assert sorted(cfg) == ["2_breakout_4096", "3_space_invaders_4096", "4_amidar_4096", "5_mixed_262144_one_gpu", "5_mixed_32768_per_gpu"]
    assert all(c["value"] > 1e6 for c in cfg.values()) and cfg["5_mixed_32768_per_gpu"]["segment_sizes"] == [10923, 10923, 10922]
    assert cfg["5_mixed_262144_one_gpu"]["segment_sizes"] == [87382, 87381, 87381] and 0.3 < cfg["5_mixed_262144_one_gpu"]["whole_step_frac"] < 1.0
    assert cfg["2_breakout_4096"]["loop"].startswith("rollout chunks") and cfg["3_space_invaders_4096"]["loop"].startswith("rollout chunks")   # (the engines' choice at 4 096 envs)
    assert j["scaling_strong"]["main"]["process"] == "own" and cfg["2_breakout_4096"]["process"] == "own"      # every arm a process of its own (run_arm)
    ss = j["scaling_strong"]
    assert ss["main"]["loop"] == "rollout chunks of 4" and ss["ring_of_8"]["loop"] == "rollout chunks of 8" and ss["ring_of_8"]["gather_every"] == 8
    assert ss["main"]["share_of_linear"] > 0.9 and ss["ring_of_8"]["share_of_linear"] > 0.9      # (measured 0.98 / 0.99-1.0; the bar here only catches a broken arm)
    assert len(j["ranks"]) == 1 and j["ranks"][0]["arch"].startswith("gfx950") and j["ranks"][0]["pci"] and j["ranks"][0]["rank"] == 0
    ap = j["agent_path"]
    for game in ("breakout", "space_invaders", "amidar", "gridworld"):
        assert ap[game]["rolled_stack"]["value"] > 1e6 and ap[game]["plane_ring"]["value"] > 0.9 * ap[game]["rolled_stack"]["value"]
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_config1"]["step_only"] > 0
    assert j["check"]["mean_score"] > 0


def test_mixed_batch_segment_sizes(oracle_lib):
    """BASELINE config 5's per-GPU share is 32 768 envs in three contiguous segments whose sizes differ by at most one
    (10 923 + 10 923 + 10 922; VERDICT r04 weak #10: 3 x 10 922 is not 32 768): MixedBatch takes one size per game, offsets and
    seeds follow the global env index, and a MixedBatch cut that way equals three engines seeded at those offsets."""
    from toybox_amd import Engine
    from toybox_amd.parallel import MixedBatch
    assert MixedBatch.split_sizes(32768, 3) == [10923, 10923, 10922] and sum(MixedBatch.split_sizes(262144 // 8, 3)) == 32768
    assert MixedBatch.split_sizes(7, 3) == [3, 2, 2] and MixedBatch.split_sizes(9, 3) == [3, 3, 3]
    games = ["breakout", "amidar", "space_invaders"]
    sizes = MixedBatch.split_sizes(20, 3)
    mb = MixedBatch(games, sizes, engine_factory=lambda g, n: Engine(g, n, lib=oracle_lib), global_offset=100)
    assert mb.n_envs == 20 and mb.sizes == [7, 7, 6] and mb.offsets == [100, 107, 114] and mb.n_per_game is None
    refs = []
    for g, n, off in zip(games, sizes, mb.offsets):
        e = Engine(g, n, lib=oracle_lib)
        e.seed(1234 + off); e.new_game()
        refs.append(e)
    for t in range(60):
        mb.step_synthetic(1337, t)
        for e, off in zip(refs, mb.offsets):
            e.step_synthetic(1337, t, env_offset=off)
    for a, b in zip(mb.engines, refs):
        for i in range(a.n_envs):
            assert bytes(a.get_state(i)) == bytes(b.get_state(i))
    assert MixedBatch(games, 4, engine_factory=lambda g, n: Engine(g, n, lib=oracle_lib)).n_per_game == 4
    with pytest.raises(ValueError):
        MixedBatch(games, [4, 4], engine_factory=lambda g, n: Engine(g, n, lib=oracle_lib))
    mb.close()
    for e in refs:
        e.close()
