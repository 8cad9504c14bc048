"""Sharding of one env batch over the GPUs of a node: one process per GPU, contiguous env ranges,
no data-path collective (envs never interact -- the reference runs one OS process per env,
/root/reference/baselines/baselines/common/vec_env/subproc_vec_env.py:49-56).  The only exchange is
the per-step all-gather of the packed {reward:i32, done:u8, lives:u8} record (8 bytes/env), which
lives BEHIND the C-ABI (tbx_gather_*, toybox_amd/csrc/gather.hip: RCCL over xGMI, resolved with
dlopen) -- no PyTorch anywhere on this path.  The ranks only have to hand the 128-byte communicator
id from rank 0 to the others; `exchange_unique_id` does that through a file, keyed so that any
launcher that sets RANK / WORLD_SIZE / MASTER_PORT (torch.distributed.run, mpirun wrappers, bench.py's
own spawner) works.

Seeds and synthetic actions are functions of the GLOBAL env index, so results do not depend on
the number of ranks.
"""
import os
import tempfile
import time

import numpy as np

from . import _abi


def shard_range(n_global, world_size, rank):
    """Contiguous partition; the first (n_global % world_size) ranks hold one extra env."""
    base, extra = divmod(int(n_global), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def unpack_records(packed):
    """uint64 records -> (reward int32, done bool, lives uint8)."""
    p = np.asarray(packed, dtype=np.uint64)
    reward = (p & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    done = ((p >> np.uint64(32)) & np.uint64(0xFF)).astype(bool)
    lives = ((p >> np.uint64(40)) & np.uint64(0xFF)).astype(np.uint8)
    return reward, done, lives


def pack_records(reward, done, lives):
    r = np.asarray(reward, dtype=np.int32).view(np.uint32).astype(np.uint64)
    d = np.asarray(done).astype(np.uint64)
    l = np.clip(np.asarray(lives, dtype=np.int64), 0, 255).astype(np.uint64)
    return r | (d << np.uint64(32)) | (l << np.uint64(40))


# ---------------------------------------------------------------------- rendezvous (no torch)

def world_from_env(env=None):
    """(rank, world_size, local_rank) as any one-process-per-GPU launcher exports them."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", env.get("RANK", "0")))


def rendezvous_key(env=None):
    """A name all ranks of ONE launch agree on and no other launch shares: an explicit TBX_RDZV_KEY, else the launcher's
    master port + run id + the pid of the common parent (all workers of torch.distributed.run / of bench.py's spawner are
    children of one process)."""
    env = os.environ if env is None else env
    if env.get("TBX_RDZV_KEY"):
        return env["TBX_RDZV_KEY"]
    return "%s_%s_%s_%d" % (env.get("MASTER_ADDR", "127.0.0.1").replace("/", "_"), env.get("MASTER_PORT", "0"),
                            env.get("TORCHELASTIC_RUN_ID", "none").replace("/", "_"), os.getppid())


_exchange_count = {}


def exchange_unique_id(rank, world, make_id, tag="", key=None, timeout=180.0, directory=None):
    """Rank 0 calls make_id() (tbx_gather_unique_id) and publishes the bytes atomically as a file; the other ranks poll for
    it.  One node, so the temp directory is shared.  Returns the id bytes on every rank.

    The file name carries the rendezvous key, the tag and the NUMBER of this exchange among the exchanges this process has
    made under that (key, tag) -- every rank makes the same exchanges in the same order, so the k-th id of a launch never
    shares a name with its (k-1)-th (a second ShardedBatch, bench.py's second communicator).  Rank 0 removes whatever a
    crashed earlier launch may have left under the name before it publishes; an explicit TBX_RDZV_KEY must still be unique
    per launch (a rank that starts before rank 0 has cleaned up could read a stale id of the same name).
    `forget_unique_id` removes the file of the last exchange once the collective init has returned."""
    if world == 1:
        return make_id()
    slot = (key or rendezvous_key(), tag)
    seq = _exchange_count.get(slot, 0)
    _exchange_count[slot] = seq + 1
    path = _id_path(tag, key, directory, seq)
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
        data = make_id()
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)
        return data
    deadline = time.monotonic() + timeout
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == _abi.GATHER_ID_BYTES:
                return data
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError("rank %d: no communicator id at %s after %.0f s" % (rank, path, timeout))
        time.sleep(0.01)


def _id_path(tag, key, directory, seq=0):
    d = directory or os.environ.get("TBX_RDZV_DIR") or tempfile.gettempdir()
    return os.path.join(d, "tbx_rccl_id_%s%s_%d" % (key or rendezvous_key(), ("_" + tag) if tag else "", seq))


def forget_unique_id(rank, tag="", key=None, directory=None):
    """rank 0: remove the file of the most recent exchange under (key, tag)"""
    if rank == 0:
        seq = _exchange_count.get((key or rendezvous_key(), tag), 1) - 1
        try:
            os.unlink(_id_path(tag, key, directory, seq))
        except OSError:
            pass


class FileWorld:
    """Barrier and max-reduction between the ranks of one node through small files in the rendezvous directory: what
    bench.py falls back to when the RCCL communicator cannot be created (the timed path has no data-path collective, so the
    measurement only needs the ranks to start and stop together)."""

    def __init__(self, rank, world, key=None, directory=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), timeout
        d = directory or os.environ.get("TBX_RDZV_DIR") or tempfile.gettempdir()
        self.base = os.path.join(d, "tbx_fw_%s" % (key or rendezvous_key()))
        self.seq = 0

    def allreduce_max(self, value):
        if self.world == 1:
            return float(value)
        self.seq += 1
        mine = "%s_%d_%d" % (self.base, self.seq, self.rank)
        tmp = mine + ".tmp"
        with open(tmp, "w") as f:
            f.write(repr(float(value)))
        os.replace(tmp, mine)
        deadline = time.monotonic() + self.timeout
        vals = []
        for r in range(self.world):
            path = "%s_%d_%d" % (self.base, self.seq, r)
            while True:
                try:
                    with open(path) as f:
                        vals.append(float(f.read()))
                    break
                except (FileNotFoundError, ValueError):
                    if time.monotonic() > deadline:
                        raise TimeoutError("rank %d: rank %d never reached step %d of the file barrier" % (self.rank, r, self.seq))
                    time.sleep(0.0005)
        if self.seq > 1:                                   # every rank wrote round seq, so every rank is done reading round seq-1
            try:
                os.unlink("%s_%d_%d" % (self.base, self.seq - 1, self.rank))
            except OSError:
                pass
        return max(vals)

    def barrier(self):
        self.allreduce_max(0.0)

    def allgather(self, obj):
        """every rank's JSON-serialisable `obj`, in rank order (the same files-in-a-directory exchange as allreduce_max)"""
        import json
        if self.world == 1:
            return [obj]
        self.seq += 1
        mine = "%s_%d_%d" % (self.base, self.seq, self.rank)
        tmp = mine + ".tmp"
        with open(tmp, "w") as f:
            f.write(json.dumps(obj))
        os.replace(tmp, mine)
        deadline = time.monotonic() + self.timeout
        vals = []
        for r in range(self.world):
            path = "%s_%d_%d" % (self.base, self.seq, r)
            while True:
                try:
                    with open(path) as f:
                        vals.append(json.loads(f.read()))
                    break
                except (FileNotFoundError, ValueError):
                    if time.monotonic() > deadline:
                        raise TimeoutError("rank %d: rank %d never reached step %d of the file exchange" % (self.rank, r, self.seq))
                    time.sleep(0.0005)
        if self.seq > 1:
            try:
                os.unlink("%s_%d_%d" % (self.base, self.seq - 1, self.rank))
            except OSError:
                pass
        return vals


class HostGather:
    """Fallback exchange when RCCL is not available: the 8-byte records are gathered on the host through a
    torch.distributed process group with a CPU backend (gloo)."""

    def __init__(self, dist, counts):
        self.dist, self.counts = dist, counts
        self.width = max(e - s for s, e in counts)

    def all_gather(self, local_records):
        import torch
        buf = torch.zeros(self.width, dtype=torch.int64)
        n = len(local_records)
        buf[:n] = torch.from_numpy(np.asarray(local_records, dtype=np.uint64).view(np.int64).copy())
        out = [torch.zeros(self.width, dtype=torch.int64) for _ in self.counts]
        self.dist.all_gather(out, buf)
        return np.stack([o.numpy().view(np.uint64) for o in out])


class ShardedBatch:
    """One rank's shard of a global env batch plus the gather of per-step records.

    engine_factory(n_local) -> Engine.  rank / world come from the launcher (world_from_env()).  The records travel through
    the engine's own tbx_gather (RCCL on the GPU) unless `host_dist` names a torch.distributed module whose CPU process
    group should carry them instead.  A single process (world == 1) needs neither: its own records ARE the gathered
    records, and no communicator is made unless the device-resident form asks for the gathered buffer.
    """

    def __init__(self, engine_factory, n_global, rank=0, world=1, seed_base=1234, host_dist=None, rdzv_tag="", rdzv_key=None,
                 gather_every=1, transport="rccl"):
        self.rank, self.world = int(rank), int(world)
        self.gather_every = max(1, int(gather_every))   # K-step record ring (TBX_OPT_GATHER_EVERY): one collective per K steps
        self.transport = transport                      # "rccl", or "host": tbx_gather over shared memory (TBX_OPT_GATHER_TRANSPORT)
        self.n_global = int(n_global)
        self.start, self.end = shard_range(n_global, self.world, self.rank)
        self.n_local = self.end - self.start
        self.counts = [shard_range(n_global, self.world, r) for r in range(self.world)]
        self.width = max(e - s for s, e in self.counts)
        self.engine = engine_factory(self.n_local)
        self.engine.seed(seed_base + self.start)      # env i gets seed_base + global index
        self.engine.new_game()
        self.host = HostGather(host_dist, self.counts) if (host_dist is not None and self.world > 1) else None
        self._rdzv = (rdzv_tag, rdzv_key)
        self.communicator = False
        if self.host is None and self.world > 1:
            self._make_communicator()

    def _make_communicator(self):
        tag, key = self._rdzv
        from . import _abi
        uid = exchange_unique_id(self.rank, self.world, self.engine.gather_unique_id, tag=tag, key=key)
        if self.transport == "host":
            self.engine.set_option(_abi.OPT_GATHER_TRANSPORT, _abi.GATHER_HOST)
        if self.gather_every > 1:
            self.engine.set_option(_abi.OPT_GATHER_EVERY, self.gather_every)   # read by gather_init
        self.engine.gather_init(self.world, self.rank, uid, records_per_rank=self.width)   # collective
        forget_unique_id(self.rank, tag=tag, key=key)
        self.communicator = True

    def _global_order(self, gathered):
        """[world][width] -> records in global env order (drops the padding of short shards)."""
        return np.concatenate([gathered[r, : e - s] for r, (s, e) in enumerate(self.counts)])

    def step_host(self, actions_global, auto_reset=True):
        """Steps the local shard with its slice of the global action vector; returns the gathered
        (reward, done, lives) over all ranks, in global env order."""
        a = np.asarray(actions_global, dtype=np.int32)[self.start:self.end]
        reward, done, lives, _ = self.engine.step(a, auto_reset=auto_reset)
        if self.host is not None:
            return unpack_records(self._global_order(self.host.all_gather(pack_records(reward, done, lives))))
        if not self.communicator:                      # one process: nothing to exchange
            return unpack_records(pack_records(reward, done, lives))
        if self.gather_every > 1:
            raise ValueError("step_host returns every step's records: use step_synthetic / gathered() with a K-step ring")
        self.engine.gather()
        return unpack_records(self._global_order(self.engine.gather_host()))

    def step_synthetic(self, action_seed, t, auto_reset=True, stream=0):
        """Device-resident form: in-kernel actions by global env index, then the asynchronous gather (results in
        TBX_BUF_GATHERED; `gathered()` fetches them)."""
        if not self.communicator and self.host is None:
            self._make_communicator()                  # world == 1 and the caller wants the gathered buffer on the device
        self.engine.step_synthetic(action_seed, t, env_offset=self.start, auto_reset=auto_reset, stream=stream)
        self.engine.gather(stream=stream)

    def gathered(self):
        """(reward, done, lives) in global env order of the last collective -- with a K-step ring a list of K such triples,
        oldest step first (the collective goes out with every K-th step_synthetic)"""
        g = self.engine.gather_host()
        if g.ndim == 3:
            return [unpack_records(self._global_order(g[:, j, :])) for j in range(g.shape[1])]
        return unpack_records(self._global_order(g))

    def max_over_ranks(self, value):
        return self.engine.gather_reduce_max(value) if self.communicator else value

    def close(self):
        self.engine.close()


class MixedBatch:
    """A batch made of several games (BASELINE config 5: Breakout + Amidar + SpaceInvaders): envs are sorted by game so
    that every launch is homogeneous -- one engine per game, each on its own HIP stream so the three step / render
    launches of a batch step overlap on the device.  Global env order: games in the order given, contiguous per game."""

    def __init__(self, games, n_per_game, device=0, seed_base=1234, engine_factory=None, global_offset=0):
        """n_per_game: one size for every segment, or one size per game (BASELINE config 5's 32 768 envs per GPU are
        10 923 + 10 923 + 10 922: `split_sizes(32768, 3)`)."""
        from .engine import Engine
        self.games = list(games)
        sizes = [int(n_per_game)] * len(self.games) if np.isscalar(n_per_game) else [int(v) for v in n_per_game]
        if len(sizes) != len(self.games) or min(sizes) < 1:
            raise ValueError("one positive segment size per game is needed")
        self.sizes = sizes
        self.n_per_game = sizes[0] if len(set(sizes)) == 1 else None
        make = engine_factory or (lambda game, n: Engine(game, n, device=device))
        self.engines = [make(g, n) for g, n in zip(self.games, sizes)]
        self.offsets = [global_offset + sum(sizes[:i]) for i in range(len(self.games))]
        for e, off in zip(self.engines, self.offsets):
            e.seed(seed_base + off)
            e.new_game()
        self.n_envs = sum(sizes)
        self.streams = None
        self.gathering = False

    @staticmethod
    def split_sizes(n_envs, n_games):
        """n_envs cut into n_games contiguous segments whose sizes differ by at most one (the first n_envs % n_games hold the
        extra env) -- the same rule as shard_range"""
        return [shard_range(n_envs, n_games, i)[1] - shard_range(n_envs, n_games, i)[0] for i in range(n_games)]

    def set_pipeline(self, value):
        """TBX_OPT_PIPELINE on every engine (those whose rasteriser reads live state ignore it): with three engines sharing one
        GPU, a step that runs beside its own previous render keeps two or three rasterisers in flight at all times."""
        from . import _abi
        for e in self.engines:
            e.set_option(_abi.OPT_PIPELINE, int(value))
        return [e.get_option(_abi.OPT_PIPELINE_ACTIVE) for e in self.engines]

    def attach_streams(self, streams):
        """One stream handle (int) per game; without it everything runs on the null stream."""
        self.streams = list(streams)

    def _stream(self, i):
        return self.streams[i] if self.streams else 0

    def gather_init(self, rank, world, rdzv_key=None, gather_every=1, transport="rccl"):
        """One communicator per game segment (every rank holds the same three segments); gather_every = K: a K-step record
        ring per segment (TBX_OPT_GATHER_EVERY); transport "rccl" or "host" (TBX_OPT_GATHER_TRANSPORT)."""
        from . import _abi
        for g, e in zip(self.games, self.engines):
            uid = exchange_unique_id(rank, world, e.gather_unique_id, tag=g, key=rdzv_key)
            e.set_option(_abi.OPT_GATHER_TRANSPORT, _abi.GATHER_HOST if transport == "host" else _abi.GATHER_RCCL)
            e.set_option(_abi.OPT_GATHER_EVERY, max(1, int(gather_every)))
            e.gather_init(world, rank, uid)
            forget_unique_id(rank, tag=g, key=rdzv_key)
        self.gathering = True

    def step_synthetic(self, action_seed, t, auto_reset=True):
        for i, (e, off) in enumerate(zip(self.engines, self.offsets)):
            e.step_synthetic(action_seed, t, env_offset=off, auto_reset=auto_reset, stream=self._stream(i))
            if self.gathering:
                e.gather(stream=self._stream(i))

    def render_device(self, channels=3):
        for i, e in enumerate(self.engines):
            e.render_device(0, channels, stream=self._stream(i))

    def render_step_synthetic(self, action_seed, t, channels=3, auto_reset=True):
        """the random-rollout loop body per segment (tbx_render_step_synthetic: one launch where the game fuses, else render then step)"""
        for i, (e, off) in enumerate(zip(self.engines, self.offsets)):
            e.render_step_synthetic(action_seed, t, channels=channels, env_offset=off, auto_reset=auto_reset, stream=self._stream(i))
            if self.gathering:
                e.gather(stream=self._stream(i))

    def step_host(self, actions_by_game, auto_reset=True):
        """actions_by_game: list of int arrays (ALE ids), one per game.  Returns per-game (reward, done, lives, score)."""
        return [e.step(a, auto_reset=auto_reset) for e, a in zip(self.engines, actions_by_game)]

    def frame_bytes(self, channels=3):
        return sum(e.n_envs * e.height * e.width * channels for e in self.engines)

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        for e in self.engines:
            e.close()
