"""Sharding of one env batch over the GPUs of a node: one process per GPU, contiguous env ranges,
no data-path collective (envs never interact -- the reference runs one OS process per env,
/root/reference/baselines/baselines/common/vec_env/subproc_vec_env.py:49-56).  The only exchange is
the per-step gather of the packed {reward:i32, done:u8, lives:u8} record (8 bytes/env), carried by
torch.distributed -- backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests.

Seeds and synthetic actions are functions of the GLOBAL env index, so results do not depend on
the number of ranks.
"""
import numpy as np


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


class ShardedBatch:
    """One rank's shard of a global env batch plus the gather of per-step records.

    engine_factory(n_local) -> Engine.  `dist` is torch.distributed (initialised by the caller) or
    None for a single process.
    """

    def __init__(self, engine_factory, n_global, dist=None, seed_base=1234):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.n_global = int(n_global)
        self.start, self.end = shard_range(n_global, self.world, self.rank)
        self.n_local = self.end - self.start
        self.engine = engine_factory(self.n_local)
        self.engine.seed(seed_base + self.start)      # env i gets seed_base + global index
        self.engine.new_game()
        self.counts = [shard_range(n_global, self.world, r) for r in range(self.world)]

    def step_host(self, actions_global, auto_reset=True):
        """Steps the local shard with its slice of the global action vector; returns the gathered
        (reward, done, lives) over all ranks, in global env order."""
        a = np.asarray(actions_global, dtype=np.int32)[self.start:self.end]
        reward, done, lives, _ = self.engine.step(a, auto_reset=auto_reset)
        local = pack_records(reward, done, lives)
        return unpack_records(self.gather(local))

    def gather(self, local_records):
        if self.dist is None or self.world == 1:
            return np.asarray(local_records, dtype=np.uint64)
        import torch
        width = max(e - s for s, e in self.counts)
        buf = torch.zeros(width, dtype=torch.int64)
        buf[: self.n_local] = torch.from_numpy(np.asarray(local_records, dtype=np.uint64).view(np.int64).copy())
        out = [torch.zeros(width, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        parts = [o.numpy().view(np.uint64)[: e - s] for o, (s, e) in zip(out, self.counts)]
        return np.concatenate(parts)


class MixedBatch:
    """A batch made of several games (BASELINE config 5: Breakout + Amidar + SpaceInvaders): envs are sorted by game so
    that every launch is homogeneous -- one engine per game, each on its own HIP stream so the three step / render
    launches of a batch step overlap on the device.  Global env order: games in the order given, contiguous per game."""

    def __init__(self, games, n_per_game, device=0, seed_base=1234, engine_factory=None, global_offset=0):
        from .engine import Engine
        self.games = list(games)
        self.n_per_game = int(n_per_game)
        make = engine_factory or (lambda game, n: Engine(game, n, device=device))
        self.engines = [make(g, self.n_per_game) for g in self.games]
        self.offsets = [global_offset + i * self.n_per_game for i in range(len(self.games))]
        for e, off in zip(self.engines, self.offsets):
            e.seed(seed_base + off)
            e.new_game()
        self.n_envs = self.n_per_game * len(self.games)
        self.streams = None

    def attach_streams(self, streams):
        """One stream handle (int) per game; without it everything runs on the null stream."""
        self.streams = list(streams)

    def _stream(self, i):
        return self.streams[i] if self.streams else 0

    def step_synthetic(self, action_seed, t, auto_reset=True):
        for i, (e, off) in enumerate(zip(self.engines, self.offsets)):
            e.step_synthetic(action_seed, t, env_offset=off, auto_reset=auto_reset, stream=self._stream(i))

    def render_device(self, channels=3):
        for i, e in enumerate(self.engines):
            e.render_device(0, channels, stream=self._stream(i))

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
