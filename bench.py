#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched game-step hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--loop auto|pair] [--gather-every K]

One "step" = one pass of the hot path over one batch: every env is stepped one frame (transition + auto-reset, actions
generated on the device by the counter-based rule of SURVEY 8d) and one uint8[N,H,W,3] frame batch is rasterised into HBM.
Workload: Breakout, 65 536 envs IN TOTAL (the metric names one 64k-env batch on 1/2/4/8 MI355X: `--scaling strong`, the
default -- rank r owns the contiguous shard r of N), env seeds 1234 + global env index.  For N > 1 there is one
process per GPU: started by any launcher that exports RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run does), or by this
script itself when it finds no RANK in its environment; the other reading (65 536 envs PER GPU, `weak`) is measured in the
same invocation with a communicator of its own and printed beside `value`, and `share_of_linear` = strong total / weak total.
The only exchange is the all-gather of the packed 8-byte {reward, done, lives} records -- tbx_gather, RCCL behind the C-ABI,
no PyTorch -- through a K-step record ring (--gather-every, default 4: one collective per 4 steps carrying all 4 steps'
records; 1 = one per step); before anything is timed one exchange is verified (own slice == TBX_BUF_PACKED, every other rank's
slice arrived) and the line says so (`rccl.verified`).  A communicator that cannot be made is a FAILED run (rc 4).

Loop forms.  The north star's loop is a random-action rollout: actions come from the device, step t+1 does not need frame t.
`fused` (default where the engine fuses: Breakout RGB / RGBA) is one tbx_render_step_synthetic per iteration -- the rasteriser
of frame t and the step to frame t+1 as ONE launch; `pair` is tbx_step_synthetic ; tbx_render_device, two launches in stream
order, the rate a policy-driven loop gets.  `chunks` (round 6; where the engine runs them: Breakout up to 32 768 envs,
SpaceInvaders up to 8 192, --rollout-chunks) is one tbx_rollout_synthetic per --gather-every steps -- step launches on a step
lane, the chunk's rasteriser launches on other internal streams (Breakout: ONE launch over the chunk's frames, chunk behind
chunk), the ring's collective queued by the call -- when the timed region is a whole number of chunks.  Whenever `value` is not the pair form in stream order, that is measured on the same engine and reported beside it
as `serialised`.

Protocol (mirroring the repeat-and-summarise shape of the reference's test/benchmark.py:119-148): an untimed
pre-roll of step-only frames so that the timed region sees mid-game states with episodes ending and auto-resets firing,
W warm-up steps, then R regions of exactly K steps, each bracketed by device-sync + rank barrier on both sides; a region's
time is the MAX over ranks; the reported value is the MEDIAN region (min / max alongside).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       -- the dominant kernel (the rasteriser; fused: rasteriser + step launch) priced against HBM bandwidth with HIP
                    events on the caller's stream: fused, around runs of 8 back-to-back launches (time / 8); two-launch form,
                    around every 8th rasteriser launch; never next to a synchronisation,
  loop           -- which loop form `value` was measured with (form, chunk_k, overlapped),
  ranks          -- one record per rank: HIP ordinal, PCI address, arch, shard (tbx_device_identity); two ranks of an RCCL run on one
                    PCI address end the run with rc 7,
  serialised     -- the two-launch loop in stream order on the same engine (when `value` is anything else),
  step_only      -- the same loop without the rasteriser (not bandwidth-bound: no roofline),
  weak / strong  -- (N > 1) the other reading, and share_of_linear,
  scaling_strong -- (N = 1) what ONE GPU does with 1/8 of the batch plus the record gather, as a share of linear (a fraction): `main`
                    (the loop form of `value`), `policy_loop` (two launches + the same ring, over `serialised`), and two launches
                    with a collective every step -- each arm in a process of its own, like the rank it stands for (`process`),
  configs        -- (N = 1, Breakout) BASELINE.json configs 2-4 (4 096 envs per game), config 5's per-GPU share (mixed 32 768 envs +
                    1-rank gather) and config 5 in full on one GPU (262 144 envs), each in a process of its own: value, serialised,
                    whole-step fraction of 8 TB/s,
  agent_path     -- (N = 1, Breakout) agent steps/s of the fused baselines wrapper stack at the headline batch size, every game,
                    rolled stack and plane ring (`--protocol agent --deepmind [--obs ring]` in short form),
  metric_version -- what `value` means (it changed between rounds 3 and 4) and which arms carry the older reading,
  rccl           -- ranks the communicator spans as RCCL reports it, ring depth, bytes per collective, library, verified,
  cpu_baseline   -- the CPU oracle (oracle/, a port: ctoybox itself cannot be built offline) on this box's host cores, a
                    bounded sample of the same 65 536-env workload, plus BASELINE config 1 (one env, one thread); rank 0, every N.
--dry-run walks the N-process launch, id exchange, barriers and teardown without touching a GPU (CPU test of the launcher).
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

# Test seams (tests/test_sharding.py walks the WHOLE N-process flow of main() -- both readings, communicators, the verified
# exchange, the JSON line -- without a GPU by putting the CPU checker and a stand-in for toybox_amd.hip here).  Never set by this
# script: without them every engine is the HIP library's and a missing GPU is an error.
ENGINE_FACTORY = None     # callable(game, n_envs, device) -> Engine
HIP_MODULE = None         # object with Stream / Event / synchronize / set_device / memcpy_dtoh


def _engine(game, n, device):
    if ENGINE_FACTORY is not None:
        return ENGINE_FACTORY(game, n, device)
    from toybox_amd import Engine
    return Engine(game, n, device=device)

# algorithmic bytes per env-step: 2*S_game + A + O + F
S_GAME = {"breakout": 72, "space_invaders": 248, "amidar": 420, "gridworld": 17}   # gridworld: player 8 + score 4 + over 4 + one cell
A_BYTES, O_BYTES = 1, 5
ACTION_SEED = 1337
SEED_BASE = 1234


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each (median reported)")
    ap.add_argument("--preroll", type=int, default=1000, help="untimed step-only frames before the warm-up (mid-game states)")
    ap.add_argument("--settle", type=int, default=40,
                    help="untimed FULL steps (step + render, the loop form about to be timed) at the end of the pre-roll, in front of the "
                         "--warmup steps: the chip comes out of the step-only pre-roll nearly idle")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="which reading is `value`.  strong (default): --envs IN TOTAL, sharded contiguously over the GPUs -- the metric "
                         "names one 64k-env batch on 1/2/4/8 GPUs; weak: --envs per GPU.  With N > 1 the other one is measured too")
    ap.add_argument("--loop", default="auto", choices=["auto", "fused", "pair"],
                    help="auto / fused: tbx_render_step_synthetic where the engine fuses (Breakout RGB / RGBA); pair: step ; render")
    ap.add_argument("--fused-overlap", default="auto", choices=["auto", "on", "off"],
                    help="TBX_OPT_FUSED_OVERLAP: consecutive fused launches on two lanes behind the device-side step ticket "
                         "(auto = the engine's choice: up to 4 096 envs; rollout chunks take precedence where both apply)")
    ap.add_argument("--rollout-chunks", default="auto", choices=["auto", "on", "off"],
                    help="TBX_OPT_ROLLOUT_CHUNKS: the fused loop as tbx_rollout_synthetic chunks of --gather-every steps (one step launch + "
                         "the chunk's rasteriser launches); auto = the engine's choice: Breakout up to 32 768 envs (from 2 048 under a record ring), SpaceInvaders up to 8 192")
    ap.add_argument("--gather-every", type=int, default=4,
                    help="K of the record ring (TBX_OPT_GATHER_EVERY): one RCCL all-gather per K steps (1 = every step)")
    ap.add_argument("--mixed-streams", type=int, default=3, choices=[1, 3],
                    help="--game mixed: one stream per segment (3: their rasterisers run side by side) or all three segments on ONE stream")
    ap.add_argument("--game", default="breakout")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--no-render", action="store_true", help="step-only mode as the main arm (no roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the step-only arm and the strong-scaling share probe")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2-5 beside the headline line (N = 1, Breakout)")
    ap.add_argument("--with-gather", action="store_true", help="run the RCCL record gather even at one rank (1-rank communicator)")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "host"],
                    help="transport of the record gather (TBX_OPT_GATHER_TRANSPORT): rccl = ncclAllGather over xGMI; host = staged through "
                         "page-locked buffers and a POSIX shared-memory segment, one node, no librccl (SURVEY 8e's fallback) -- the line then "
                         "says `rccl: null, gather: {transport: host}`")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank uses device 0 (with --gather host: the whole N-process flow on a one-GPU box; RCCL refuses two ranks "
                         "of one communicator on one device)")
    ap.add_argument("--strict-rccl", action="store_true",
                    help="N > 1: a run whose RCCL communicator cannot be made or verified FAILS (rc 4) instead of falling back to --gather host")
    ap.add_argument("--allow-no-gather", action="store_true",
                    help="N > 1 only: if no RCCL communicator can be made, run without the per-step gather (file barrier) instead of failing")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1, 2, 3], help="TBX_OPT_PIPELINE of the main arm (1 = engine's choice)")
    ap.add_argument("--in-process-arms", action="store_true",
                    help="run BASELINE configs 2-5 and the strong-scaling probe in this process instead of one process per arm")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / barrier walk-through without any GPU call")
    ap.add_argument("--gym", action="store_true", help="reference protocol: also the env.step() arm (test/benchmark.py:83-97)")
    ap.add_argument("--reps", type=int, default=30, help="reference protocol: repetitions (mean and s.e.m. reported)")
    ap.add_argument("--protocol", default="batch", choices=["batch", "reference", "agent", "host"],
                    help="'reference' = the raw single-env loop of the reference's harness (test/benchmark.py:44-58)")
    ap.add_argument("--deepmind", action="store_true",
                    help="agent protocol: also EpisodicLife + FireReset + NoopReset(30) + episode monitor (wrap_deepmind)")
    ap.add_argument("--obs", default="stack", choices=["stack", "ring"],
                    help="agent protocol: 'stack' = the rolled uint8[N,84,84,4] on the device (VecFrameStack's array); 'ring' = one new "
                         "plane per step into a ring of the last 4 (tbx_agent_config_t::new_plane = 2: the reference's data flow)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the CPU arm at the headline batch")
    return ap.parse_args()


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU boxes report 256
    logical CPUs but run the job under a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


# ---------------------------------------------------------------------------------------------- CPU arms (the checker, timed)

def _oracle_lib():
    from toybox_amd import _abi
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    _abi.bind(lib)
    return lib


def cpu_baseline(segments, channels, target_seconds):
    """The CPU oracle on all usable host cores at the headline batch size.  segments = [(game, n_envs, global offset)]; every
    segment is held as chunks of 4 096 envs (one frame buffer per game is reused, so host memory stays ~1.5 GB instead of full
    frames for every env), step + render with auto-reset and the same action rule, for about `target_seconds` of wall time
    (step count calibrated from a short probe)."""
    from toybox_amd import Engine
    lib = _oracle_lib()
    if lib is None:
        return None
    cores = usable_cores()
    os.environ["TBX_ORACLE_THREADS"] = str(cores)
    engines, frames, total = [], {}, 0
    for game, n_seg, off in segments:
        for c0 in range(0, n_seg, 4096):                # (the last chunk of a ragged segment is shorter)
            chunk = min(4096, n_seg - c0)
            e = Engine(game, chunk, lib=lib)
            e.seed(SEED_BASE + off + c0)
            e.new_game()
            engines.append((e, off + c0))
            total += chunk
            if game not in frames:
                frames[game] = np.empty((min(4096, n_seg), e.height, e.width, channels), np.uint8)

    def run(t_from, count):
        t0 = time.perf_counter()
        for t in range(t_from, t_from + count):
            for e, off in engines:
                e.step_synthetic(ACTION_SEED, t, env_offset=off)
                e.render_device(frames[e.game].ctypes.data, channels)
        return time.perf_counter() - t0

    run(0, 1)                                           # warm-up (thread pool, page faults)
    probe = run(1, 2) / 2 + 1e-9                        # calibration
    steps = int(max(3, min(20000, target_seconds / probe)))
    dt = run(3, steps)
    for e, _ in engines:
        e.close()
    what = " + ".join("%s %d" % (g, n) for g, n, _ in segments)
    return {"value": total * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%s envs (%d in chunks of <= 4096) step+render(%dch) x %d steps, OpenMP static partition over envs, %.1f s" %
                      (what, total, channels, steps, dt)}


def cpu_config1(game, channels):
    """BASELINE config 1: one env, one thread, 1000 random-action steps on the CPU path (here the oracle: the Rust core cannot
    be built offline), step-only like the reference's harness loop (test/benchmark.py:50-56) and with the RGB frame."""
    from toybox_amd import Engine
    lib = _oracle_lib()
    if lib is None:
        return None
    os.environ["TBX_ORACLE_THREADS"] = "1"
    out = {"unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "%s, 1 env, 1000 steps, one ctypes call per step" % game}
    for key, render in (("step_only", False), ("step_render", True)):
        e = Engine(game, 1, lib=lib)
        e.seed(SEED_BASE)
        e.new_game()
        for t in range(100):
            e.step_synthetic(ACTION_SEED, t)
        t0 = time.perf_counter()
        for t in range(100, 1100):
            e.step_synthetic(ACTION_SEED, t)
            if render:
                e.render_device(0, channels)
        out[key] = 1000.0 / (time.perf_counter() - t0)
        e.close()
    return out


class quiet_stdout:
    """librccl prints a version banner on STDOUT when a communicator is created; this script's stdout carries exactly one JSON
    line, so fd 1 points at stderr while a communicator is being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        try:                                   # the banner sits in libc's stdio buffer when stdout is a pipe: push it out
            import ctypes                      # while fd 1 still points at stderr, not after the JSON line at exit
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


# ---------------------------------------------------------------------------------------------- timed regions

class Region:
    """R regions of K steps: device-sync + rank barrier on both sides of each, MAX over ranks per region."""

    def __init__(self, sync, rank_barrier, rank_max):
        self.sync, self.rank_barrier, self.rank_max = sync, rank_barrier, rank_max

    def run(self, one_step, t0_index, K, R, on_region=None, many=None):
        """many(t, K): the K steps of a region as one call of the loop object (rollout chunks) instead of K calls of one_step"""
        times = []
        t = t0_index
        for _ in range(R):
            self.sync(); self.rank_barrier()
            if on_region is not None:
                on_region()
            w0 = time.perf_counter()
            if many is not None:
                many(t, K)
            else:
                for i in range(K):
                    one_step(t + i)
            self.sync(); self.rank_barrier()
            times.append(self.rank_max(time.perf_counter() - w0))
            t += K
        return times, t


def summarize(times, K):
    in_order = [1000.0 * x / K for x in times]
    ms = sorted(in_order)
    return {"n": len(ms), "ms_per_step_median": statistics.median(ms), "ms_per_step_min": ms[0], "ms_per_step_max": ms[-1],
            "ms_per_step_in_run_order": [round(v, 5) for v in in_order]}


# ---------------------------------------------------------------------------------------------- other protocols

def mixed_reading(args, world, rank, local_rank, envs, K, Wm, R, gather, cpu_seconds=None):
    """BASELINE config 5: Breakout + Amidar + SpaceInvaders, `envs` envs per GPU in three contiguous segments whose sizes
    differ by at most one (32 768 = 10 923 + 10 923 + 10 922), three homogeneous launches per phase on three streams, one
    record gather per segment.  Returns the JSON object on rank 0 (None elsewhere)."""
    from toybox_amd import hip
    from toybox_amd.parallel import MixedBatch
    games = ["breakout", "amidar", "space_invaders"]
    sizes = MixedBatch.split_sizes(envs, len(games))
    mb = MixedBatch(games, sizes, device=local_rank, global_offset=rank * envs)
    streams = [hip.Stream() for _ in range(1 if args.mixed_streams == 1 else len(games))]
    mb.attach_streams([streams[i % len(streams)].ptr for i in range(len(games))])
    if gather:
        with quiet_stdout():
            mb.gather_init(rank, world, gather_every=args.gather_every, transport=GATHER_TRANSPORT)
    C = args.channels
    render = not args.no_render
    fused = render and args.loop != "pair" and C >= 3
    # three engines on three streams already keep two or three rasterisers in flight; the pipelined mode on top of that was
    # measured slower (0.93-1.01 ms per step against 0.83-0.91: its internal streams, the three callers' streams and three
    # communication streams then share the runtime's few hardware queues): off unless asked for with --pipeline 2 / 3
    pipe = mb.set_pipeline(0 if args.pipeline == 1 else args.pipeline)
    lead = mb.engines[0]
    reg = Region(hip.synchronize, (lambda: lead.gather_reduce_max(0.0)) if gather else (lambda: None),
                 (lambda v: lead.gather_reduce_max(v)) if gather else (lambda v: v))

    def one(t):
        if fused:
            mb.render_step_synthetic(ACTION_SEED, t, C)       # per segment: one launch where the game fuses, else render ; step
            return
        mb.step_synthetic(ACTION_SEED, t)
        if render:
            mb.render_device(C)

    for e, off in zip(mb.engines, mb.offsets):
        for t in range(args.preroll):
            e.step_synthetic(ACTION_SEED, t, env_offset=off, auto_reset=True)
    for t in range(SETTLE + Wm):
        one(args.preroll + t)
    times, _ = reg.run(one, args.preroll + SETTLE + Wm, K, R)
    mb.sync()
    out = None
    if rank == 0:
        total = world * mb.n_envs
        fb = mb.frame_bytes(C) if render else 0
        rep = summarize(times, K)
        ms = rep["ms_per_step_median"]
        out = {"metric": "env steps/sec (whole node), mixed Breakout+Amidar+SpaceInvaders batch", "value": total / (ms * 1e-3),
               "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": ms, "repeats": rep,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+int32", "data": "synthetic",
               "pipeline": {"option": args.pipeline, "resolved_per_game": dict(zip(games, pipe))},
               "loop": {"form": "fused" if fused else "pair", "fused_per_game": dict(zip(games, [e.get_option(_abi_mod().OPT_RENDER_STEP_FUSED) for e in mb.engines]))},
               "config": {"workload": "mixed batch, %d envs/GPU = %s (breakout, amidar, space_invaders), %s, three streams%s"
                                      % (mb.n_envs, " + ".join(str(v) for v in sizes), "step + RGB render" if render else "step-only",
                                         (", RCCL gather of 8 B/env records, one collective per %d steps" % max(1, args.gather_every)) if gather else ""),
                          "envs_per_gpu": mb.n_envs, "envs_total": total, "segment_sizes": sizes},
               "roofline": ({"bound": "hbm", "kernel": "the three rasterisers together (whole-step time, not per kernel)",
                             "achieved": fb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "algorithmic_bytes_per_step": fb,
                             "per_kernel": "profiles/ (rocprofv3 kernel trace of this command: the three rasterisers' own durations and "
                                           "whether they overlap)"} if render else None),
               "rccl": ({"transport": GATHER_TRANSPORT, "nranks": lead.gather_nranks(), "gather_bytes_per_step": 8 * mb.n_envs * world, "lib": lead.gather_library(),
                         "gather_every": lead.gather_every(), "communicators": len(games)} if gather else None)}
        if cpu_seconds:
            try:
                out["cpu_baseline"] = cpu_baseline([(g, n, off) for g, n, off in zip(games, sizes, mb.offsets)], C, cpu_seconds)
            except Exception as ex:
                out["cpu_baseline"] = {"error": repr(ex)}
    mb.close()
    for st in streams:
        st.close()
    return out


def bench_mixed(args, world, rank, local_rank):
    out = mixed_reading(args, world, rank, local_rank, args.envs, args.steps, args.warmup, args.repeats, world > 1 or args.with_gather,
                        cpu_seconds=None if args.no_cpu_baseline else args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0


def _abi_mod():
    from toybox_amd import _abi
    return _abi


def _mean_sem(xs):
    xs = np.asarray(xs, dtype=np.float64)
    sem = float(xs.std(ddof=1) / np.sqrt(len(xs))) if len(xs) > 1 else 0.0
    return float(xs.mean()), sem


def bench_reference_protocol(args):
    """The reference's own harness, test/benchmark.py: the RAW arm (:44-58: one env, action = legal[i % len(legal)],
    new_game() when game_over() else apply_ale_action(move), no rendering) and with --gym the ENV arm (:83-97: gym-style env,
    random agent, obs, reward, done, _ = env.step(action), reset when done -- the observation is rendered and copied to the host
    every step); FPS = steps / elapsed per repetition, --reps repetitions, mean and s.e.m. as :119-148 print them, and the
    harness's "slowdown" of the env arm against the raw arm.  One FFI round trip per frame: on a GPU this measures latency,
    not throughput; the CPU oracle runs the same loops beside it."""
    from toybox_amd import Engine
    from toybox_amd import toybox as tbmod
    from toybox_amd.envs import ENV_IDS
    from toybox_amd.toybox import Toybox
    nsteps = args.steps if args.steps != 200 else 10000      # (200 is this script's batch-protocol default; the harness uses 10 000)
    reps = max(1, args.reps)
    env_id = {"breakout": "BreakoutToyboxNoFrameskip-v4", "amidar": "AmidarToyboxNoFrameskip-v4",
              "space_invaders": "SpaceInvadersToyboxNoFrameskip-v4"}.get(args.game)

    def raw_loop(tb):
        actions = tb.get_legal_action_set()
        t0 = time.perf_counter()
        for i in range(nsteps):
            move = actions[i % len(actions)]
            if tb.game_over():
                tb.new_game()
            else:
                tb.apply_ale_action(move)
        return nsteps / (time.perf_counter() - t0)

    def env_loop(env, rng):
        n_act = env.action_space.n
        env.reset()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            obs, reward, done, _ = env.step(int(rng.integers(n_act)))
            if done:
                env.reset()
        return nsteps / (time.perf_counter() - t0)

    def arms(label):
        res = {}
        with Toybox(args.game) as tb:
            raw_loop(tb)                                        # warm-up (first launch, resident kernel start)
            res["raw"] = [raw_loop(tb) for _ in range(reps)]
        if args.gym and env_id:
            env = ENV_IDS[env_id]()
            rng = np.random.default_rng(0)
            env_loop(env, rng)
            res["gym"] = [env_loop(env, rng) for _ in range(reps)]
            env.close()
        return res

    gpu = arms("gpu")
    mean, sem = _mean_sem(gpu["raw"])
    out = {"metric": "raw single-env steps/sec, reference harness protocol (test/benchmark.py:44-58)", "unit": "env-steps/s",
           "value": mean, "sem": sem, "reps": reps, "n_gpus": 1, "steps": nsteps, "warmup": 1, "ms_per_step": 1000.0 / mean,
           "higher_is_better": True, "vs_baseline": None, "data": "synthetic", "scaling": "weak", "dtype": "f64" if args.game == "breakout" else "int32",
           "config": {"workload": "%s single env, cycling legal actions, new_game on game over, no render; %d reps x %d steps"
                                  % (args.game, reps, nsteps)}}
    if "gym" in gpu:
        gm, gs = _mean_sem(gpu["gym"])
        out["gym"] = {"value": gm, "sem": gs, "unit": "env-steps/s", "slowdown_vs_raw": (mean - gm) / mean,
                      "workload": "%s, random agent, env.step() returns the (H, W, 1) gray observation on the host every step" % env_id}
    lib = _oracle_lib()
    if lib is not None and not args.no_cpu_baseline:
        tbmod.set_engine_factory(lambda game, n: Engine(game, n, lib=lib))
        cpu = arms("cpu")
        tbmod.set_engine_factory(None)
        cm, cs = _mean_sem(cpu["raw"])
        out["cpu_baseline"] = {"value": cm, "sem": cs, "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": "same raw loop over the CPU oracle, %d reps x %d steps" % (reps, nsteps)}
        if "gym" in cpu:
            gm, gs = _mean_sem(cpu["gym"])
            out["cpu_baseline"]["gym"] = {"value": gm, "sem": gs, "slowdown_vs_raw": (cm - gm) / cm}
    print(json.dumps(out), flush=True)
    return 0


def bench_agent_protocol(args):
    """agent steps/s of the fused wrapper stack (skip 4, 84x84 gray, stack 4, clipped reward): one agent
    step = 4 game frames + 2 gray renders + max/warp/stack; only 28 KB per env leave the pass."""
    from toybox_amd import Engine, hip
    n, K, Wm = args.envs, args.steps, args.warmup
    eng = Engine(args.game, n, device=0)
    eng.seed(SEED_BASE)
    dm = bool(args.deepmind)
    eng.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=dm, fire_reset=dm,
                   noop_max=30 if dm else 0, noop_seed=2024, new_plane=2 if args.obs == "ring" else 0)
    eng.agent_reset()
    stream = hip.Stream()
    for t in range(Wm):
        eng.agent_step_synthetic(ACTION_SEED, t, stream=stream.ptr)
    hip.synchronize()
    t0 = time.perf_counter()
    for t in range(Wm, Wm + K):
        eng.agent_step_synthetic(ACTION_SEED, t, stream=stream.ptr)
    hip.synchronize()
    dt = time.perf_counter() - t0
    eng.sync()
    out = {"metric": "agent steps/sec (skip-4, 84x84x4 obs), %s" % args.game, "value": n * K / dt, "unit": "agent-steps/s",
           "env_frames_per_s": 4 * n * K / dt, "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": 1000 * dt / K,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "%s fused %sMaxAndSkip(4)+WarpFrame(84)+ClipReward+%s, %d envs, device actions"
                                  % (args.game, "NoopReset(30)+EpisodicLife+FireReset+Monitor+" if dm else "",
                                     "ring of the last 4 planes (the stack as LazyFrames)" if args.obs == "ring" else "FrameStack(4)", n),
                      "obs": args.obs}}
    print(json.dumps(out), flush=True)
    eng.close()
    return 0


def bench_host_protocol(args):
    """The PCIe-inclusive rate: what a caller on the HOST side of the boundary gets, the way the reference's consumers sit
    (ToyboxBaseEnv / VecEnv hand every frame to numpy: envs/atari/base.py:109, vec_env/__init__.py:63-74).  One step = actions
    from a host array, the batch step, the observation of every env in host memory -- through the VecEnv classes' step().
    Arms: the RGB frames through ToyboxVecEnv (rotating page-locked pool, the default; a fresh pageable array per step, the
    contract as the reference spells it) and the agent pipeline through ToyboxPreprocVecEnv in its three layouts: whole
    uint8[N,84,84,4] stacks from the device (default pool / fresh pageable arrays), ONE new plane per env and step with the stack
    kept on the host as planes (the reference's data flow: subproc_vec_env.py:63-74 + vec_frame_stack.py:17-30), and the same
    transfer rolled into a real array on the host (tbx_host_stack_push, threaded).  Never `value` of the headline: that one is measured with everything resident
    in HBM."""
    import numpy as np
    from toybox_amd.envs import ToyboxPreprocVecEnv, ToyboxVecEnv
    n, K, Wm = args.envs, args.steps, args.warmup
    rng = np.random.default_rng(ACTION_SEED)
    arms = {}

    def run(env, unit, bytes_per_step, steps):
        na = env.action_space.n
        env.reset()
        acts = [rng.integers(0, na, n) for _ in range(8)]
        for t in range(Wm):
            env.step(acts[t % 8])
        t0 = time.perf_counter()
        for t in range(steps):
            env.step(acts[t % 8])
        dt = time.perf_counter() - t0
        env.close()
        return {"value": n * steps / dt, "unit": unit, "ms_per_step": 1000 * dt / steps, "pcie_GB_per_s": bytes_per_step * steps / dt / 1e9}

    frame_bytes = None
    for name, pool in (("frames_pinned_pool", 2), ("frames_pageable_fresh_array", 0)):
        env = ToyboxVecEnv(args.game, n, grayscale=False, seed=SEED_BASE, obs_pool=pool)
        frame_bytes = int(np.prod(env.observation_space.shape))
        arms[name] = run(env, "env-steps/s", n * frame_bytes, K if pool else max(3, K // 4))
    px = 84 * 84
    for name, layout, pool, per_env, steps in (("agent_planes_pinned_ring", "planes", 2, px, 4 * K),
                                               ("agent_device_stack_pinned_pool", "device_stack", 2, 4 * px, 2 * K),
                                               ("agent_device_stack_pageable_fresh_array", "device_stack", 0, 4 * px, K),
                                               ("agent_host_stack_native_roll", "host_stack", 2, px, max(3, K // 4))):
        env = ToyboxPreprocVecEnv(args.game, n, seed=SEED_BASE, obs_layout=layout, obs_pool=pool)
        arms[name] = run(env, "agent-steps/s", n * per_env, steps)
    best = arms["frames_pinned_pool"]
    out = {"metric": "env steps/sec INCLUDING the PCIe transfer of every frame to the host, %s" % args.game, "value": best["value"],
           "unit": "env-steps/s", "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": best["ms_per_step"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic", "arms": arms,
           "config": {"workload": "%s ToyboxVecEnv.step(host actions) -> RGB frames in host memory, %d envs, %d B per frame; "
                                  "ToyboxPreprocVecEnv.step -> 84x84x4 observations" % (args.game, n, frame_bytes)}}
    print(json.dumps(out), flush=True)
    return 0


# ---------------------------------------------------------------------------------------------- launcher

def spawn_ranks(args):
    """No RANK in the environment and --gpus N > 1: start the N ranks ourselves (before anything touches a GPU), pass rank
    0's line through, exit with the worst return code.  A rank that dies takes the others with it (they would wait for it in
    the communicator set-up or at the next barrier for ever)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    key = "bench_%d_%d" % (os.getpid(), int(time.time() * 1e3))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TBX_RDZV_KEY=key)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def dry_run(args, rank, world):
    """The N-process part of a run without a GPU: every rank finds its place (RANK / WORLD_SIZE), receives the 128-byte
    communicator id from rank 0 through the rendezvous file, checks that all ranks hold the SAME id, walks R regions of K
    barrier-bracketed "steps" with a max-over-ranks reduction, and rank 0 prints the JSON line.  What it cannot cover is RCCL
    itself."""
    import hashlib
    from toybox_amd.parallel import FileWorld, exchange_unique_id, forget_unique_id, shard_range
    from toybox_amd._abi import GATHER_ID_BYTES
    uid = exchange_unique_id(rank, world, lambda: os.urandom(GATHER_ID_BYTES))
    fw = FileWorld(rank, world)
    h = int.from_bytes(hashlib.sha256(uid).digest()[:6], "little")
    same = fw.allreduce_max(float(h)) == float(h) and fw.allreduce_max(-float(h)) == -float(h)
    forget_unique_id(rank)
    if not same:
        print("bench.py --dry-run: rank %d holds another communicator id than its peers" % rank, file=sys.stderr)
        return 5
    if args.scaling == "strong":
        start, end = shard_range(args.envs, world, rank)
        n_total = args.envs
    else:
        start, end = rank * args.envs, (rank + 1) * args.envs
        n_total = world * args.envs
    covered = fw.allreduce_max(float(end))                    # the last rank's end is the whole batch
    reg = Region(lambda: None, fw.barrier, fw.allreduce_max)
    times, _ = reg.run(lambda t: time.sleep(0.0002 * (1 + rank)), 0, args.steps, max(1, args.repeats))
    rep = summarize(times, args.steps)
    if rank == 0:
        ms = rep["ms_per_step_median"]
        print(json.dumps({"metric": "dry run (no GPU work)", "dry_run": True, "value": n_total / (ms * 1e-3), "unit": "env-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": 0, "ms_per_step": ms, "repeats": rep,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "data": "none",
                          "config": {"workload": "launcher walk-through", "envs_total": n_total, "envs_covered": int(covered),
                                     "envs_per_gpu": end - start},
                          "rccl": None, "id_exchange": "ok: %d ranks hold the same %d-byte id" % (world, GATHER_ID_BYTES)}), flush=True)
    return 0 if int(covered) == n_total else 6


class Loop:
    """The timed loop over one engine.  Two forms of one iteration ("step" of the bench contract = one frame stepped AND one
    frame rasterised for every env):
      pair   tbx_step_synthetic ; [tbx_gather] ; tbx_render_device   -- two launches in stream order (what a policy loop does)
      fused  tbx_render_step_synthetic ; [tbx_gather]                -- the frame of the current state and the step to the
             next one in ONE launch where the rasteriser reads step-written records (Breakout RGB); random rollouts only
    Timing of the dominant kernel (roofline.avg_launch_ms) with HIP events on the caller's stream, never next to a device
    synchronisation (the first SKIP launches of a region run against an idle memory system and are left out):
      fused  RUNS of RUN back-to-back launches between two events -- the events of a region form a chain, each one the end of a
             run and the start of the next -- divided by RUN: what one launch costs in the steady state of the loop;
      pair   the rasteriser launches are separated by step kernels, so every RUN-th rasteriser launch is bracketed on its own."""

    RUN = 8         # launches per event pair (an event costs the stream a few microseconds: around every launch that was 1 % of
                    # the 65 536-env step and 5 % of the 8 192-env one)
    SKIP = 2        # launches after a region's opening synchronisation that no event pair covers

    def __init__(self, eng, hip, stream, start, channels, gather, render, K, R, fused=False, rollout=None):
        from toybox_amd import _abi
        self.eng, self.sp, self.start, self.C, self.gather, self.render = eng, stream.ptr, start, channels, gather, render
        self.fused = bool(fused and render)
        # a random rollout (actions do not depend on the frame): the arm may run as rollout chunks even where the single call is two
        # launches (SpaceInvaders); default: wherever the fused call is asked for
        self.rollout = self.fused if rollout is None else bool(rollout and render)
        # overlapped fused launches (TBX_OPT_FUSED_OVERLAP): the caller's stream joins lazily -- when an address is asked for.  A mark
        # there is a join + a timing event + (next call) a fence for the lane: measured at 8 192 envs, a mark every 8 launches cost the
        # loop 4-6 % (0.1715 against 0.1576 ms per step with the K = 4 ring) -- so an overlapped region is ONE span, two marks
        self.overlapped = self.fused and eng.get_option(_abi.OPT_FUSED_OVERLAP_ACTIVE) == 1
        self._frame_id = _abi.BUF_FRAME
        self.K = int(K)
        self.run = max(1, min(self.RUN, self.K - self.SKIP)) if self.K > 1 else 1
        self.skip = max(0, min(self.SKIP, self.K - self.run))
        if self.overlapped and self.K > 1:
            self.run = self.K - self.skip
        # rollout chunks (tbx_rollout_synthetic / TBX_OPT_ROLLOUT_CHUNKS): the fused loop in chunks of `chunk_k` steps -- one step launch
        # + the chunk's rasteriser launches (one over its chunk_k x n frames, or one per frame on two lanes) per call, the ring's collective queued by the call itself.  Only when every phase of the
        # loop is a whole number of chunks (a K-step ring cannot be left partly filled)
        self.chunk_k = 0
        self._rollout_ids = (_abi.BUF_ROLLOUT_FRAMES,)
        per_region = 0
        if render and K > 0:
            per_region = ((self.K - self.skip) // self.run + 1) if self.fused else 2 * ((self.K - self.skip + self.run - 1) // self.run)
        self.pool = [hip.Event() for _ in range(per_region * max(0, R))]
        self.next_ev = 0
        self.armed = False          # nothing is timed until arm() is called
        self.spans = []             # (start event, end event, launches covered)
        self.chain = None
        self.c = 0

    def arm(self):
        self.armed = True
        self.next_ev = 0
        self.spans = []

    def begin_region(self):
        self.c = 0
        self.chain = None

    def _mark(self):
        if self.next_ev >= len(self.pool):
            return None
        ev = self.pool[self.next_ev]
        self.next_ev += 1
        if self.overlapped:
            self.eng.device_buffer(self._frame_id)     # "where is the last frame": the stream now waits for the launch that wrote it
        ev.record(self.sp)
        return ev

    def full_step(self, t):
        e, sp = self.eng, self.sp
        c = self.c
        self.c += 1
        timing = self.armed and self.render and c >= self.skip
        if not self.fused:
            e.step_synthetic(ACTION_SEED, t, env_offset=self.start, auto_reset=True, stream=sp)
            if self.gather:
                e.gather(stream=sp)            # on the engine's communication stream: overlaps with the rasteriser below
            if self.render:
                a = self._mark() if timing and (c - self.skip) % self.run == 0 else None
                e.render_device(0, self.C, stream=sp)
                if a is not None:
                    b = self._mark()
                    if b is not None:
                        self.spans.append((a, b, 1))
            return
        if timing and c == self.skip:
            self.chain = self._mark()
        e.render_step_synthetic(ACTION_SEED, t, channels=self.C, env_offset=self.start, auto_reset=True, stream=sp)
        if self.gather:
            e.gather(stream=sp)                # the records of the step that rode in the launch above
        if timing and self.chain is not None and (c + 1 - self.skip) % self.run == 0:
            b = self._mark()
            if b is not None:
                self.spans.append((self.chain, b, self.run))
            self.chain = b

    def use_chunks(self, k, phases):
        """switch to the chunk form if the engine runs it overlapped and every phase length in `phases` is a multiple of k"""
        from toybox_amd import _abi
        if not self.rollout or k < 2 or any(p % k for p in phases):
            return False
        if self.eng.get_option(_abi.OPT_ROLLOUT_CHUNKS_ACTIVE) != 1:
            return False
        if self.gather and self.eng.gather_every() != k:
            return False
        self.chunk_k = k
        self.overlapped = True
        return True

    def many_steps(self, t, count):
        """`count` steps in the chunk form (count is a multiple of chunk_k); one event span per armed region: behind the first chunk
        to behind the last one"""
        e, sp, k = self.eng, self.sp, self.chunk_k
        first = None
        for c in range(count // k):
            e.rollout_synthetic(ACTION_SEED, t + c * k, k, channels=self.C, env_offset=self.start, auto_reset=True, stream=sp)
            if self.armed and self.render and c == 0 and count > k:
                e.device_buffer(self._rollout_ids[0])              # (lazy join: the stream now waits for this chunk's launches)
                first = self._mark_raw()
        if first is not None:
            e.device_buffer(self._rollout_ids[0])
            last = self._mark_raw()
            if last is not None:
                self.spans.append((first, last, count - k))

    def _mark_raw(self):
        if self.next_ev >= len(self.pool):
            return None
        ev = self.pool[self.next_ev]
        self.next_ev += 1
        ev.record(self.sp)
        return ev

    def step_only(self, t):
        self.eng.step_synthetic(ACTION_SEED, t, env_offset=self.start, auto_reset=True, stream=self.sp)
        if self.gather:
            self.eng.gather(stream=self.sp)

    def launch_ms(self):
        """per-launch times of the timed spans (call after a device synchronisation): list of ms, launches covered"""
        per = [a.elapsed_ms(b) / k for a, b, k in self.spans]
        return per, sum(k for _, _, k in self.spans)

    def close(self):
        for ev in self.pool:
            ev.close()
        self.pool = []


SETTLE = 40     # untimed full steps of the loop form about to be timed, in front of its W warm-up steps (--settle)


def timed_arm(eng, hip, reg, stream, start, C, gather, render, pipeline, t, K, Wm, R, fused=False, rollout=None):
    """SETTLE untimed steps + W warm-up steps + R regions of K steps with TBX_OPT_PIPELINE = pipeline (pair form) or the fused
    call.  Returns (summary, launch timing dict or None, resolved pipeline mode, next t).
    Why the settle steps: the pre-roll is 1 000 ten-microsecond step kernels, and the first dozen rasteriser launches behind it
    run 2-19 % long (kernel trace of `--steps 20 --warmup 5`: 1 224, 1 296, 1 425, 1 430, 1 411, 1 348 ... 1 204 us, clocks and
    memory system coming up from a nearly idle chip); with a 5-step warm-up they landed in the first timed region.  They belong
    to the untimed pre-roll, like the 1 000 frames before them; the W warm-up steps and the K timed steps are the caller's."""
    from toybox_amd import _abi
    eng.set_option(_abi.OPT_PIPELINE, 0 if fused else pipeline)
    eng.set_option(_abi.OPT_FUSED_OVERLAP, {"auto": _abi.FUSED_OVERLAP_AUTO, "on": _abi.FUSED_OVERLAP_ON, "off": _abi.FUSED_OVERLAP_OFF}[FUSED_OVERLAP])
    mode = eng.get_option(_abi.OPT_PIPELINE_ACTIVE)
    eng.set_option(_abi.OPT_ROLLOUT_CHUNKS, {"auto": _abi.ROLLOUT_CHUNKS_AUTO, "on": _abi.ROLLOUT_CHUNKS_ON, "off": _abi.ROLLOUT_CHUNKS_OFF}[ROLLOUT_CHUNKS])
    loop = Loop(eng, hip, stream, start, C, gather, render, K, R, fused=fused, rollout=rollout)
    chunked = loop.use_chunks(CHUNK_K, (K,))
    if chunked:
        warm = -(-(SETTLE + Wm) // CHUNK_K) * CHUNK_K       # untimed: rounded UP to whole chunks (a record ring cannot be left partly filled)
        loop.many_steps(t, warm)
        t += warm
    else:
        for _ in range(SETTLE + Wm):
            loop.full_step(t)
            t += 1
    hip.synchronize()
    loop.arm()
    times, t = reg.run(loop.full_step, t, K, R, on_region=loop.begin_region, many=loop.many_steps if chunked else None)
    per, covered = loop.launch_ms()
    launch = None
    if per:
        launch = {"avg_ms": float(np.mean(per)), "median_ms": float(np.median(per)), "min_ms": float(min(per)), "max_ms": float(max(per)),
                  "spans": len(per), "launches": covered, "launches_per_span": loop.run if loop.fused else 1,
                  "skipped_after_sync": loop.skip, "overlapped": loop.overlapped, "chunk_k": loop.chunk_k}
    LAST_LOOP_FORM["chunk_k"] = loop.chunk_k
    LAST_LOOP_FORM["overlapped"] = loop.overlapped
    loop.close()
    return summarize(times, K), launch, mode, t


def launch_timing_note(fused, mode):
    if fused:
        return ("HIP events on the caller's stream around RUNS of %d back-to-back launches (chained: one event ends a run and starts the "
                "next), time / %d; the first %d launches after a region's opening synchronisation are not covered" % (Loop.RUN, Loop.RUN, Loop.SKIP))
    return ("HIP events on the caller's stream around every %dth rasteriser launch of the timed regions (step kernels sit between "
            "rasteriser launches in this loop form), never one of the first %d after a synchronisation" % (Loop.RUN, Loop.SKIP) +
            ("" if mode != 3 else "; launches overlap in this mode, so this is the time from one launch's end to the next one's end "
                                  "(what a launch costs in steady state), not a kernel's own duration"))


PIPELINE_NOTE = {0: "off: every call in stream order", 2: "the step runs beside the previous frame's rasteriser (internal step stream, "
                 "two sets of render records and step outputs)", 3: "the step runs beside the previous frame's rasteriser and consecutive "
                 "rasteriser launches alternate between two internal streams and two frame buffers"}
FUSED_NOTE = ("tbx_render_step_synthetic: the rasteriser of frame t and the batch step to frame t+1 are ONE launch (the step's blocks "
              "in front of the rasteriser's, other records buffer); random-action rollouts only -- `serialised` is the two-launch loop")


GATHER_TRANSPORT = "rccl"      # --gather
FUSED_OVERLAP = "auto"         # --fused-overlap
ROLLOUT_CHUNKS = "auto"        # --rollout-chunks
CHUNK_K = 4                    # --gather-every: the ring depth is the chunk length
LAST_LOOP_FORM = {"chunk_k": 0, "overlapped": False}    # what timed_arm's loop resolved to (read right after the call)
CHUNK_NOTE = ("tbx_rollout_synthetic: the fused loop in chunks of %d steps -- per chunk ONE step launch on an internal stream (every env %d "
              "frames with its state in registers: %d render records, the %d step records straight into the record ring, the state once) and "
              "the rasteriser launches of its %d frames, dependent on that step launch alone -- ONE launch over the chunk's frames, chunk behind "
              "chunk on a second internal stream (Breakout under a record ring or above 4 096 envs), or a launch per frame alternating between "
              "two; the next chunk's step launch runs beside this chunk's rasterisers and the ring's collective waits for the step launch only "
              "(TBX_OPT_ROLLOUT_CHUNKS; the engines' choice: Breakout up to 32 768 envs, SpaceInvaders up to 8 192).  avg_launch_ms: the "
              "time per frame between the ends of the first and the last chunk of a region -- what a frame costs in steady state -- not a "
              "kernel's own duration")
OVERLAP_NOTE = ("consecutive fused launches alternate between two internal streams, output sets and frame buffers; launch N+1 is ordered "
                "behind the STEP BLOCKS of launch N only (a device-side counter they bump once their agent-scope stores are out, awaited by "
                "a one-wave kernel in front of launch N+1), so it ramps up while launch N still paints (TBX_OPT_FUSED_OVERLAP; the engine's "
                "choice up to 4 096 envs).  avg_launch_ms in this mode is the time from one launch's end to the next one's end -- what a "
                "launch costs in steady state -- not a kernel's own duration")


def make_communicator(eng, rank, world, width, gather_every, tag, transport=None):
    """tbx_gather_init over `world` ranks (id from rank 0 through the rendezvous file), K-step record ring if asked.  Returns the
    `rccl` object of the JSON line; raises when no communicator over `world` ranks comes out of it."""
    from toybox_amd import _abi
    from toybox_amd.parallel import exchange_unique_id, forget_unique_id
    transport = transport or GATHER_TRANSPORT
    if os.environ.get("TBX_BENCH_NO_RCCL") and transport == "rccl":
        raise RuntimeError("disabled by TBX_BENCH_NO_RCCL")
    eng.set_option(_abi.OPT_GATHER_TRANSPORT, _abi.GATHER_HOST if transport == "host" else _abi.GATHER_RCCL)
    eng.set_option(_abi.OPT_GATHER_EVERY, max(1, gather_every))
    with quiet_stdout():
        uid = exchange_unique_id(rank, world, eng.gather_unique_id, tag=tag)
        eng.gather_init(world, rank, uid, records_per_rank=width)   # collective (ncclCommInitRank)
    forget_unique_id(rank, tag=tag)
    K = eng.gather_every()
    rccl = {"transport": transport, "nranks": eng.gather_nranks(), "records_per_rank": width, "gather_every": K,
            "gather_bytes_per_collective": 8 * width * world * K, "gather_bytes_per_step": 8 * width * world, "lib": eng.gather_library()}
    if rccl["nranks"] != world:
        raise RuntimeError("the communicator spans %d ranks, not %d" % (rccl["nranks"], world))
    return rccl


def verify_gather(eng, hip, rank, world, n_local, shard_sizes, start):
    """One real exchange before anything is timed: K steps (K = ring depth) with the gather queued, then the gathered block is
    read back -- this rank's slice must equal its own TBX_BUF_PACKED records, and in every OTHER rank's slice every env's
    `lives` byte must be non-zero (fresh games: a slice that never arrived reads zero).  Returns True or raises."""
    from toybox_amd import _abi
    K = eng.gather_every()
    for j in range(K):
        eng.step_synthetic(ACTION_SEED, j, env_offset=start, auto_reset=True)
        eng.gather()
    got = eng.gather_host().reshape(world, K, -1)
    mine = np.empty(n_local, np.uint64)
    p, _ = eng.device_buffer(_abi.BUF_PACKED)
    hip.synchronize()
    hip.memcpy_dtoh(mine, p, 8 * n_local)
    if not np.array_equal(got[rank, K - 1, :n_local], mine):
        raise RuntimeError("rank %d: its own slice of the gathered records differs from TBX_BUF_PACKED" % rank)
    for r in range(world):
        lives = (got[r, :, :shard_sizes[r]] >> np.uint64(40)) & np.uint64(0xFF)
        if not lives.all():
            raise RuntimeError("rank %d: the slice of rank %d did not arrive (zero lives fields)" % (rank, r))
    return True


def run_reading(args, hip, game, rank, world, local_rank, scaling, tag, with_extras, extras=("serialised", "step_only")):
    """One reading of the metric on this rank: engine for its shard, communicator (N > 1 or --with-gather) with one verified
    exchange, pre-roll, the timed arm (fused where the engine fuses, unless --loop pair), and with_extras the serialised
    two-launch loop and the step-only loop beside it (`extras` names which).  Returns a dict (rank 0 assembles the line) or an
    int return code."""
    from toybox_amd import Engine, _abi
    from toybox_amd.parallel import FileWorld, rendezvous_key, shard_range
    if scaling == "strong":
        spans = [shard_range(args.envs, world, r) for r in range(world)]
        n_total = args.envs
    else:
        spans = [(r * args.envs, (r + 1) * args.envs) for r in range(world)]
        n_total = world * args.envs
    start, end = spans[rank]
    sizes = [e - s for s, e in spans]
    width = max(sizes)
    n = end - start
    eng = _engine(game, n, local_rank)
    eng.seed(SEED_BASE + start)            # env i of this rank: seed 1234 + global index
    eng.new_game()
    # which device every rank drives, as the engine itself reports it (hipGetDeviceProperties behind tbx_device_identity), gathered
    # through the rendezvous directory: the line shows that N ranks drove N distinct GPUs, and an RCCL run in which two ranks report
    # the same PCI address is refused below (VERDICT r05 #4b; per-rank placement as baselines/common/cmd_util.py:31 seeds per rank)
    ident = dict(eng.device_identity(), rank=rank, local_rank=local_rank, envs=[start, end])
    ranks = FileWorld(rank, world, key=rendezvous_key() + "_ident_" + tag).allgather(ident) if world > 1 else [ident]
    H, W, C = eng.height, eng.width, args.channels
    render = not args.no_render
    gather = world > 1 or args.with_gather
    gather_note, rccl, fw = None, None, None
    if gather:
        def attempt(transport, tag_):
            try:
                c = make_communicator(eng, rank, world, width, args.gather_every, tag_, transport=transport)
                c["verified"] = verify_gather(eng, hip, rank, world, n, sizes, start)
                eng.new_game()             # (the verification stepped K frames)
                return c, None
            except Exception as ex:
                return None, (str(ex).splitlines()[0][:200] if str(ex) else repr(ex))
        pcis = [r_["pci"] for r_ in ranks if r_.get("pci")]
        if GATHER_TRANSPORT == "rccl" and world > 1 and len(set(pcis)) != len(pcis) and not (args.one_device or os.environ.get("TBX_BENCH_ONE_DEVICE")):
            print("bench.py: rank %d: two ranks of an RCCL run drive the same device (%s) -- one process per GPU is the contract"
                  % (rank, ", ".join("rank %d: %s" % (r_["rank"], r_["pci"]) for r_ in ranks)), file=sys.stderr)
            return 7
        rccl, msg = attempt(GATHER_TRANSPORT, tag)
        # the ranks agree on the outcome through the rendezvous directory (a communicator that came up on some ranks only is no
        # communicator): if RCCL failed anywhere, EVERY rank falls back to the host transport ("a host-staged gather
        # is the fallback if RCCL is missing") and the line says so -- `rccl: null`, `gather.fallback_from_rccl` -- instead of
        # there being no line at all; --strict-rccl keeps the failed run
        failed_somewhere = bool(msg)
        if world > 1:
            from toybox_amd.parallel import rendezvous_key
            agree = FileWorld(rank, world, key=rendezvous_key() + "_agree_" + tag)
            failed_somewhere = agree.allreduce_max(1.0 if msg else 0.0) > 0.0
        if failed_somewhere and GATHER_TRANSPORT == "rccl" and world > 1 and not args.strict_rccl:
            first_msg = msg or "another rank could not make or verify its RCCL communicator"
            print("bench.py: rank %d: no verified RCCL gather over %d ranks (%s): falling back to the host transport"
                  % (rank, world, first_msg), file=sys.stderr)
            rccl, msg = attempt("host", tag + "_host")
            if rccl is not None:
                rccl["fallback_from_rccl"] = first_msg
            failed_somewhere = agree.allreduce_max(1.0 if msg else 0.0) > 0.0
        if failed_somewhere:
            msg = msg or "another rank could not make or verify its communicator"
            if world > 1 and not args.allow_no_gather:
                # the north star's 8-GPU number INCLUDES the collective: a run that cannot make it is a failed run
                print("bench.py: rank %d: no verified record gather over %d ranks (%s); pass --allow-no-gather to measure the shards "
                      "without the per-step gather" % (rank, world, msg), file=sys.stderr)
                return 4
            gather_note = "communicator unavailable (%s): no per-step gather, file barrier between ranks" % msg
            print("bench.py: " + gather_note, file=sys.stderr)
            gather, rccl = False, None
            fw = FileWorld(rank, world) if world > 1 else None
    stream = hip.Stream()
    K, Wm, R = args.steps, args.warmup, max(1, args.repeats)
    if gather:
        reg = Region(hip.synchronize, lambda: eng.gather_reduce_max(0.0), lambda v: eng.gather_reduce_max(v))
    elif fw is not None:
        reg = Region(hip.synchronize, fw.barrier, fw.allreduce_max)
    else:
        reg = Region(hip.synchronize, lambda: None, lambda v: v)

    t = 0
    for _ in range(args.preroll):          # untimed: bring the batch to mid-game states (episodes end, auto-resets fire)
        eng.step_synthetic(ACTION_SEED, t, env_offset=start, auto_reset=True, stream=stream.ptr)
        t += 1
    fused = render and args.loop != "pair" and C >= 3 and eng.get_option(_abi.OPT_RENDER_STEP_FUSED) == 1
    rep, launch, mode, t = timed_arm(eng, hip, reg, stream, start, C, gather, render, args.pipeline, t, K, Wm, R, fused=fused,
                                     rollout=render and args.loop != "pair")
    res = {"n": n, "n_total": n_total, "start": start, "H": H, "W": W, "C": C, "render": render, "gather": gather, "rccl": rccl,
           "gather_note": gather_note, "rep": rep, "launch": launch, "mode": mode, "fused": fused, "steps": K, "extras": {},
           "overlapped": LAST_LOOP_FORM["overlapped"], "chunk_k": LAST_LOOP_FORM["chunk_k"], "ranks": ranks}
    frame_bytes = H * W * C if render else 0
    if with_extras and "serialised" in extras and (fused or mode != 0 or LAST_LOOP_FORM["chunk_k"]):
        # the same engine, two launches per frame in stream order: what a policy-driven loop (actions computed from the frame) gets
        srep, sl, _, t = timed_arm(eng, hip, reg, stream, start, C, gather, render, 0, t, K, Wm, R, fused=False)
        sms = srep["ms_per_step_median"]
        s_ms = sl["avg_ms"] if sl else None
        res["extras"]["serialised"] = {"value": n_total / (sms * 1e-3), "unit": "env-steps/s", "ms_per_step": sms, "repeats": srep,
                                       "avg_launch_ms": s_ms,
                                       "roofline_frac": (n * frame_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if s_ms else None,
                                       "whole_step_frac": n * frame_bytes / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "note": "tbx_step_synthetic ; tbx_render_device in stream order (TBX_OPT_PIPELINE = 0): the rate of a loop whose "
                                               "actions depend on the frame"}
        eng.set_option(_abi.OPT_PIPELINE, args.pipeline)
    if render and with_extras and "step_only" in extras:
        loop = Loop(eng, hip, stream, start, C, gather, False, 0, 0)
        so_times, t = reg.run(loop.step_only, t, K, R)
        so = summarize(so_times, K)
        res["extras"]["step_only"] = {"value": n_total / (so["ms_per_step_median"] * 1e-3), "unit": "env-steps/s",
                                      "ms_per_step": so["ms_per_step_median"], "repeats": so,
                                      "note": "same loop without the rasteriser; latency / issue bound, no roofline"}
    # sanity: the rollout really played (scores move, lives are lost, episodes end)
    eng.sync()
    score, lives, level, over = eng.scalars()
    res["check"] = {"mean_score": float(score.mean()), "mean_lives": float(lives.mean()), "max_level": int(level.max()), "frames_played": t}
    eng.close()
    if t >= 300 and not (res["check"]["mean_score"] > 0):
        print("bench.py: the rollout did not play (mean score %.3f after %d frames)" % (res["check"]["mean_score"], t), file=sys.stderr)
        return 3
    return res


def main():
    global SETTLE, GATHER_TRANSPORT, FUSED_OVERLAP, ROLLOUT_CHUNKS, CHUNK_K
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with "invalid argument" under the
    # legacy mode); the GPU boxes export it already -- set before anything loads the HIP runtime, for launchers that do not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get(ARM_ENV):                        # the child side of run_arm: one arm of the line in a process of its own
        return arm_main(json.loads(os.environ.pop(ARM_ENV)))
    args = parse()
    SETTLE = max(0, args.settle)
    GATHER_TRANSPORT = args.gather
    FUSED_OVERLAP = args.fused_overlap
    ROLLOUT_CHUNKS = args.rollout_chunks
    CHUNK_K = max(1, args.gather_every)
    if args.protocol == "reference":
        return bench_reference_protocol(args)
    if args.protocol == "agent":
        return bench_agent_protocol(args)
    if args.protocol == "host":
        return bench_host_protocol(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)

    from toybox_amd.parallel import world_from_env
    rank, world, local_rank = world_from_env()
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run(args, rank, world)

    if HIP_MODULE is not None:
        hip = HIP_MODULE
    else:
        from toybox_amd import hip
    if args.one_device or os.environ.get("TBX_BENCH_ONE_DEVICE"):     # every rank on device 0 (the N > 1 flow on a 1-GPU box)
        local_rank = 0
    hip.set_device(local_rank)

    game = args.game
    if game == "mixed":
        return bench_mixed(args, world, rank, local_rank)

    # The metric names ONE batch -- "Breakout 64k-env batch, 1/2/4/8 MI355X" -- so `value` is the reading with --envs IN TOTAL,
    # sharded contiguously over the ranks (env i -> GPU i / (N / G)); with N > 1 the other reading (--envs per GPU,
    # `weak`) is measured in the same invocation and printed beside it, each with its own verified communicator.
    main_res = run_reading(args, hip, game, rank, world, local_rank, args.scaling, "main", not args.no_extras)
    if isinstance(main_res, int):
        return main_res
    other = None
    if world > 1 and not args.no_extras:
        other_scaling = "weak" if args.scaling == "strong" else "strong"
        other = run_reading(args, hip, game, rank, world, local_rank, other_scaling, "other", False)
        if isinstance(other, int):
            return other

    if rank == 0:
        r = main_res
        n, n_total, H, W, C, render, gather = r["n"], r["n_total"], r["H"], r["W"], r["C"], r["render"], r["gather"]
        rep, mode, fused = r["rep"], r["mode"], r["fused"]
        frame_bytes = H * W * C if render else 0
        bytes_per_step = 2 * S_GAME[game] + A_BYTES + O_BYTES + frame_bytes
        ms = rep["ms_per_step_median"]
        K_ring = r["rccl"]["gather_every"] if r["rccl"] else 1
        host_gather = (r["rccl"] or {}).get("transport") == "host"
        out = {
            "metric": "env steps/sec (whole node), Breakout 64k-env batch" if game == "breakout" else "env steps/sec (whole node), %s" % game,
            "value": n_total / (ms * 1e-3),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "repeats": rep,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64" if game == "breakout" else "int32",
            "data": "synthetic",
            "config": {
                "workload": "%s %s, %d envs %s, uniform random legal actions generated on device "
                            "(splitmix64 counter rule, seed 1337), env seeds 1234+global index, auto-reset on done, "
                            "untimed pre-roll of %d step-only frames + %d full steps before the warm-up"
                            % (game, "step + %dx%dx%d uint8 frame render" % (H, W, C) if render else "step-only",
                               args.envs, "per GPU" if args.scaling == "weak" else "in total", args.preroll, SETTLE),
                "envs_per_gpu": n, "envs_total": n_total, "frame_hwc": [H, W, C] if render else None,
                "parallelism": parallelism_note(world, K_ring, fused, gather, host_gather, r),
                "algorithmic_bytes_per_env_step": bytes_per_step,
            },
            "loop": {"form": "chunks" if r["chunk_k"] else "fused" if fused else "pair",
                     "what": FUSED_NOTE if fused else "tbx_step_synthetic ; tbx_render_device, two launches per frame",
                     "overlapped": r["overlapped"], "chunk_k": r["chunk_k"],
                     "overlap": (CHUNK_NOTE % ((r["chunk_k"],) * 5)) if r["chunk_k"] else OVERLAP_NOTE if r["overlapped"] else
                     ("stream order (--fused-overlap %s; the engine runs rollout chunks up to 8 192 envs and overlaps single fused launches up to 4 096 -- see configs and scaling_strong)" % FUSED_OVERLAP if fused else None)},
            "pipeline": {"option": args.pipeline, "resolved": mode, "what": PIPELINE_NOTE.get(mode),
                         "applies_to": "the two-launch loop form only; see `serialised`"},
            "rccl": r["rccl"] if (r["rccl"] or {}).get("transport") == "rccl" else None,
            "gather": r["rccl"],
            "ranks": r["ranks"],
        }
        if render:
            out["roofline"] = roofline_object(game, n, frame_bytes, C, fused, mode, r["launch"])
        else:
            out["roofline"] = None
        out["metric_version"] = METRIC_VERSION
        out.update(r["extras"])
        out["check"] = r["check"]
        if other is not None:
            oms = other["rep"]["ms_per_step_median"]
            key = "weak" if args.scaling == "strong" else "strong"
            out[key] = {"value": other["n_total"] / (oms * 1e-3), "unit": "env-steps/s", "ms_per_step": oms, "repeats": other["rep"],
                        "envs_per_gpu": other["n"], "envs_total": other["n_total"],
                        "rccl": other["rccl"] if (other["rccl"] or {}).get("transport") == "rccl" else None, "gather": other["rccl"],
                        "loop": "fused" if other["fused"] else "pair",
                        "note": "the other reading of the metric, measured in the same invocation with its own communicator"}
            s_, w_ = (out, out[key]) if args.scaling == "strong" else (out[key], out)
            # strong total over weak total = what N GPUs make of ONE batch against N x a full batch each: the share of linear scaling
            out["share_of_linear"] = s_["value"] / w_["value"]
    # ---- the arms that only one GPU's worth of a run carries (every rank walks the same code; ranks > 0 have nothing to do)
    single = world == 1 and ENGINE_FACTORY is None
    if rank == 0 and single and not args.no_extras and main_res["n"] >= 16384:
        try:
            out["scaling_strong"] = strong_share_probe(args, game, main_res["C"], main_res["n"], out["value"],
                                                       (out.get("serialised") or {}).get("value") or out["value"], hip)
        except Exception as ex:
            out["scaling_strong"] = {"error": repr(ex)}
    if rank == 0 and single and not args.no_extras and not args.no_configs and game == "breakout" and main_res["render"]:
        # BASELINE.json configs 2-5 beside the headline (config 1 is cpu_config1 below): every arm from one invocation, as the
        # reference's harness prints all of its arms from one run (test/benchmark.py:119-166)
        try:
            out["configs"] = baseline_configs(args, hip)
        except Exception as ex:
            out["configs"] = {"error": repr(ex)}
        # ... and the learner-side pipeline of every game, so that its rates are driver-timed too
        try:
            out["agent_path"] = agent_path_rates(hip, main_res["n"])
        except Exception as ex:
            out["agent_path"] = {"error": repr(ex)}
    if rank == 0 and not args.no_cpu_baseline:
        # the CPU path "in the same run" (north star), also for N > 1: rank 0's host cores, a bounded sample of the same batch,
        # after every GPU region (the other ranks have nothing left to do and leave)
        try:
            out["cpu_baseline"] = cpu_baseline([(game, main_res["n_total"], 0)], main_res["C"], args.cpu_seconds)
            out["cpu_config1"] = cpu_config1(game, main_res["C"])
        except Exception as ex:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(ex)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0


# `value` changed meaning between rounds 3 and 4: the same `bench.py --gpus N` line is comparable across rounds only
# under the same version.  1 = rounds 1-3: --envs PER GPU (weak), two launches per step, one collective per step.
# 2 = round 4 on: --envs IN TOTAL (strong; identical at N = 1), the fused launch where the engine fuses, K = 4 record ring.
# The version-1 reading of the same engine is still in every line: `serialised` (two launches) and, for N > 1, `weak`.
METRIC_VERSION = {"version": 2, "value_is": "strong reading (envs in total), loop form the engine offers a random rollout, one collective per "
                                            "--gather-every steps",
                  "version_1_arms_in_this_line": {"two launches per step": "serialised", "envs per GPU (N > 1)": "weak",
                                                  "a collective every step (N = 1 probe)": "scaling_strong.pair_gather_every_step"}}


def parallelism_note(world, K_ring, fused, gather, host_gather, r):
    if not gather:
        return ("env-sharded x%d, no collective (%s)" % (world, r["gather_note"])) if r["gather_note"] else "single GPU"
    if host_gather:
        fb = (r["rccl"] or {}).get("fallback_from_rccl")
        head = ("HOST-STAGED FALLBACK (RCCL failed: %s) -- this `value` understates the design by the whole overlap of the collective: " % fb) if fb else ""
        return head + ("env-sharded x%d, HOST-STAGED all-gather of 8 B/env records behind the C-ABI (tbx_gather over a POSIX shared-memory "
                       "segment, no RCCL: every collective blocks the calling thread until the step has finished and all ranks have "
                       "exchanged), one per %d step(s)" % (world, K_ring))
    return "env-sharded x%d, RCCL all-gather of 8 B/env records behind the C-ABI (tbx_gather), %s" % (world, gather_overlap_note(K_ring, fused))


def gather_overlap_note(K_ring, fused):
    if K_ring > 1:
        return ("K-step record ring: one collective per %d steps on the engine's communication stream, overlapping the launches of "
                "the following steps (the step that re-opens a ring, 2 K steps later, waits for it)" % K_ring)
    if fused:
        return ("one collective per step, queued behind each fused launch and waited for by the next one: NOT overlapped in this loop "
                "form (the ring, --gather-every > 1, or --loop pair overlaps it)")
    return "one collective per step on the engine's communication stream, overlapped with the rasteriser launch that follows the step"


def csrc_fingerprint():
    """sha256 (16 hex digits) over the kernel sources and the header as they lie in this tree -- what a static measurement
    (profiles/traffic.json) is tied to: a figure measured on other sources is reported as stale.  The same rule
    is in scripts/summarize_profile.py, which writes the figure."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "toybox_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "toybox_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def roofline_object(game, n, frame_bytes, C, fused, mode, launch):
    """The dominant kernel (rasteriser; fused: rasteriser + step launch) against HBM bandwidth: algorithmic frame bytes of one
    launch over the event-timed launch time of the loop's steady state."""
    if not launch:
        return None
    ms = launch["avg_ms"]
    achieved = n * frame_bytes / (ms * 1e-3) / 1e9            # GB/s
    traffic, source, stale, measured_on = None, None, None, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            rec = json.load(open(tp)).get("%s_render_%dch_%d%s" % (game, C, n, "_fused" if fused else ""))
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
                source = "profiles/traffic.json (static: rocprofv3 PMC pass %s, not measured in this run)" % rec.get("source", "")
                measured_on = rec.get("csrc_sha16")
                # stale: the kernel sources of this tree are not the ones the counters were read on (unknown for entries of rounds <= 5)
                stale = (measured_on != csrc_fingerprint()) if measured_on else None
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": "%s render (%d ch)%s" % (game, C, " + step, one launch" if fused else ""),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": source, "traffic_stale": stale, "traffic_measured_on_csrc": measured_on,
            "algorithmic_bytes_per_launch": n * frame_bytes, "avg_launch_ms": ms, "median_launch_ms": launch["median_ms"],
            "min_launch_ms": launch["min_ms"], "max_launch_ms": launch["max_ms"], "launches_timed": launch["launches"],
            "event_spans": launch["spans"], "timing": launch_timing_note(fused, mode)}


def agent_path_rates(hip, n, steps=40, warmup=12):
    """agent steps/s of the fused wrapper stack (NoopReset(30) + MaxAndSkip(4) + Monitor + EpisodicLife + FireReset + WarpFrame(84) +
    ClipReward + FrameStack(4); device-generated actions) at the headline batch size, per game, with the rolled uint8[N,84,84,4] stack
    on the device and with the ring of the last 4 planes instead (tbx_agent_config_t::new_plane = 2): `bench.py --protocol agent
    --deepmind [--obs ring]` in short form (one region of `steps` agent steps after `warmup`)."""
    from toybox_amd import Engine
    rates = {"unit": "agent-steps/s", "envs": n, "steps": steps, "agent_step": "4 game frames + observation, every baselines wrapper"}
    stream = hip.Stream()
    for game in ("breakout", "space_invaders", "amidar", "gridworld"):
        rates[game] = {}
        for obs, mode in (("rolled_stack", 0), ("plane_ring", 2)):
            eng = Engine(game, n, device=0)
            eng.seed(SEED_BASE)
            eng.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=True, fire_reset=True, noop_max=30,
                           noop_seed=2024, new_plane=mode)
            eng.agent_reset()
            for t in range(warmup):
                eng.agent_step_synthetic(ACTION_SEED, t, stream=stream.ptr)
            hip.synchronize()
            t0 = time.perf_counter()
            for t in range(warmup, warmup + steps):
                eng.agent_step_synthetic(ACTION_SEED, t, stream=stream.ptr)
            hip.synchronize()
            dt = time.perf_counter() - t0
            eng.close()
            rates[game][obs] = {"value": n * steps / dt, "ms_per_step": 1000 * dt / steps}
    stream.close()
    return rates


ARM_ENV = "TBX_BENCH_ARM"


def run_arm(kind, key, args, extra, hip, timeout=900):
    """One arm of the line -- a BASELINE config, an arm of the strong-scaling probe -- in a process of its OWN, as that workload's own
    run would be (and as a rank of an N-GPU run is: it never hosted a 65 536-env engine first).  Measured: the 8 192-env chunk loop
    runs 0.154-0.156 ms per step in a fresh process and 0.162-0.165 as the second engine of the process that ran the headline batch
    (stream order: 0.160 in both); the small-batch rows of the line were 5-20 % behind their own invocations for the same reason
    (profiles/r06_experiments.txt 4).  --in-process-arms (or a failing child) runs the arm here instead and says so."""
    if not getattr(args, "in_process_arms", False):
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TBX_RDZV_KEY")}
        env[ARM_ENV] = json.dumps({"kind": kind, "key": key, "args": vars(args), "extra": extra})
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=timeout)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                out = json.loads(lines[-1])
                out["process"] = "own"
                return out
            note = "child rc %d: %s" % (p.returncode, (p.stderr or "").strip().splitlines()[-1][:160] if (p.stderr or "").strip() else "no line")
        except Exception as ex:
            note = repr(ex)[:160]
    else:
        note = "--in-process-arms"
    out = ARMS[kind](args, key, extra, hip)
    out["process"] = "shared with the headline arm (%s)" % note
    return out


def arm_main(payload):
    """the child side of run_arm"""
    global SETTLE, GATHER_TRANSPORT, FUSED_OVERLAP, ROLLOUT_CHUNKS, CHUNK_K
    args = argparse.Namespace(**payload["args"])
    args.in_process_arms = True
    SETTLE = max(0, args.settle)
    GATHER_TRANSPORT, FUSED_OVERLAP, ROLLOUT_CHUNKS, CHUNK_K = args.gather, args.fused_overlap, args.rollout_chunks, max(1, args.gather_every)
    from toybox_amd import hip
    hip.set_device(0)
    print(json.dumps(ARMS[payload["kind"]](args, payload["key"], payload["extra"], hip)), flush=True)
    return 0


def config_arm(args, key, extra, hip):
    """BASELINE config 2, 3 or 4: 4 096 envs of one game, the protocol of the headline but regions of >= 200 steps"""
    g = extra["game"]
    a = argparse.Namespace(**vars(args))
    a.envs, a.steps, a.warmup, a.repeats, a.with_gather, a.no_render, a.scaling = 4096, max(args.steps, 200), 20, 5, False, False, "strong"
    r = run_reading(a, hip, g, 0, 1, 0, "strong", "cfg_" + g, True, extras=("serialised",))
    if isinstance(r, int):
        return {"error": "rc %d" % r}
    ms = r["rep"]["ms_per_step_median"]
    fb = r["H"] * r["W"] * r["C"]
    e = {"value": r["n"] / (ms * 1e-3), "unit": "env-steps/s", "ms_per_step": ms, "steps": a.steps, "repeats": r["rep"]["n"],
         "ms_per_step_min_max": [r["rep"]["ms_per_step_min"], r["rep"]["ms_per_step_max"]],
         "loop": ("rollout chunks of %d" % r["chunk_k"]) if r["chunk_k"] else ("fused, overlapped launches" if r["overlapped"] else "fused") if r["fused"] else "pair",
         "pipeline_resolved": r["mode"],
         "whole_step_frac": r["n"] * fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         # (overlapped launches -- TBX_OPT_PIPELINE 3, TBX_OPT_FUSED_OVERLAP, rollout chunks: an event pair on the caller's stream does not
         # bracket a kernel; it is the steady-state period of a launch, printed as such)
         "kernel_frac": (r["n"] * fb / (r["launch"]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if r["launch"] and r["mode"] == 0 else None,
         "avg_launch_ms": r["launch"]["avg_ms"] if r["launch"] and r["mode"] == 0 else None,
         "launch_timing": "end-to-end period of overlapped launches" if r["overlapped"] else "events around launches in stream order",
         "frame_hwc": [r["H"], r["W"], r["C"]]}
    sr = r["extras"].get("serialised")
    e["serialised"] = ({"value": sr["value"], "ms_per_step": sr["ms_per_step"], "whole_step_frac": sr["whole_step_frac"],
                        "kernel_frac": sr["roofline_frac"]} if sr else "= value (the main arm is the two-launch loop in stream order)")
    return e


def mixed_arm(args, key, extra, hip):
    """BASELINE config 5: its per-GPU share (32 768 envs) or the whole mixed batch (262 144 envs) on one GPU, 1-rank record gather"""
    a = argparse.Namespace(**vars(args))
    a.no_render = False
    m = mixed_reading(a, 1, 0, 0, extra["envs"], extra["steps"], extra["warmup"], extra["repeats"], True)
    return {"value": m["value"], "unit": "env-steps/s" + (" per GPU" if extra["envs"] == 32768 else ""), "ms_per_step": m["ms_per_step"], "steps": m["steps"],
            "ms_per_step_min_max": [m["repeats"]["ms_per_step_min"], m["repeats"]["ms_per_step_max"]],
            "segment_sizes": m["config"]["segment_sizes"], "loop": m["loop"], "rccl": m["rccl"],
            "whole_step_frac": m["roofline"]["frac"], "frame_bytes_per_step": m["roofline"]["algorithmic_bytes_per_step"], "note": extra["note"]}


def baseline_configs(args, hip):
    """BASELINE.json configs 2, 3, 4 (4 096 envs of Breakout / SpaceInvaders / Amidar on one GPU), the per-GPU share of config 5
    (mixed 32 768 envs = 10 923 + 10 923 + 10 922 with the 1-rank record gather) and config 5 in full on ONE GPU (262 144 envs =
    87 382 + 87 381 + 87 381, ~38 GB of frames: the denominator any later 8-GPU line of this config needs), each in a
    process of its own (run_arm).  `whole_step_frac` = frame bytes of the batch / ms_per_step / 8 TB/s."""
    cfgs = {}
    for key, g in (("2_breakout_4096", "breakout"), ("3_space_invaders_4096", "space_invaders"), ("4_amidar_4096", "amidar")):
        try:
            cfgs[key] = run_arm("config", key, args, {"game": g}, hip)
        except Exception as ex:
            cfgs[key] = {"error": repr(ex)[:300]}
    for key, extra in (("5_mixed_32768_per_gpu", {"envs": 32768, "steps": max(args.steps, 50), "warmup": 10, "repeats": 5,
                                                  "note": "the per-GPU share of BASELINE config 5 (262 144 envs over 8 GPUs) on ONE GPU with the record gather "
                                                          "queued over a 1-rank communicator: launch cost of the collective, no wire time"}),
                       ("5_mixed_262144_one_gpu", {"envs": 262144, "steps": min(max(args.steps, 10), 30), "warmup": 5, "repeats": 3,
                                                   "note": "BASELINE config 5 in full on one GPU: three homogeneous segments on three streams, the record gather "
                                                           "queued over a 1-rank communicator per segment (K = %d ring); 8 GPUs would each take a 32 768-env "
                                                           "share of it (5_mixed_32768_per_gpu)" % max(1, args.gather_every)})):
        try:
            cfgs[key] = run_arm("mixed", key, args, extra, hip)
        except Exception as ex:
            cfgs[key] = {"error": repr(ex)[:300]}
    return cfgs


def probe_arm(args, key, extra, hip):
    """one arm of the strong-scaling probe: n envs (1/8 of the batch) with the record gather on"""
    from toybox_amd import Engine, _abi
    global CHUNK_K
    game, C, n, every, want_fused = extra["game"], extra["C"], extra["n"], extra["every"], extra["fused"]
    K = max(args.steps, 200)
    chunk_k_before, CHUNK_K = CHUNK_K, max(1, every)          # (a rollout chunk under a record ring is the ring's K steps)
    eng = Engine(game, n, device=0)
    eng.seed(SEED_BASE)
    eng.new_game()
    eng.set_option(_abi.OPT_GATHER_EVERY, max(1, every))
    eng.set_option(_abi.OPT_GATHER_TRANSPORT, _abi.GATHER_HOST if GATHER_TRANSPORT == "host" else _abi.GATHER_RCCL)
    with quiet_stdout():
        eng.gather_init(1, 0, eng.gather_unique_id())
    st = hip.Stream()
    for t in range(args.preroll):
        eng.step_synthetic(ACTION_SEED, t, auto_reset=True, stream=st.ptr)
    reg = Region(hip.synchronize, lambda: eng.gather_reduce_max(0.0), lambda v: eng.gather_reduce_max(v))
    fused = want_fused and C >= 3 and eng.get_option(_abi.OPT_RENDER_STEP_FUSED) == 1
    rep, launch, mode, _ = timed_arm(eng, hip, reg, st, 0, C, True, True, args.pipeline, args.preroll, K, 20, 5, fused=fused)
    v = n / (rep["ms_per_step_median"] * 1e-3)
    out = {"value": v, "ms_per_step": rep["ms_per_step_median"], "repeats": rep,
           "loop": ("rollout chunks of %d" % LAST_LOOP_FORM["chunk_k"]) if LAST_LOOP_FORM["chunk_k"] else
                   ("fused, overlapped launches" if LAST_LOOP_FORM["overlapped"] else "fused") if fused else "pair",
           "gather_every": eng.gather_every(), "pipeline_resolved": mode, "avg_launch_ms": launch["avg_ms"] if launch else None}
    eng.close()
    st.close()
    CHUNK_K = chunk_k_before
    return out


def strong_share_probe(args, game, C, n_single, single_value, single_pair_value, hip):
    """What ONE GPU of an 8-GPU run of the SAME batch would do: n/8 envs with the record gather on (1-rank communicator: launch
    and stream-hop cost of the collective, no wire time), each arm in a process of its own like the rank it stands for (run_arm).
    share_of_linear = that rate over the single-GPU rate of the whole batch (a fraction: 1.0 = eight GPUs are eight times one).
    Measured for the loop form and ring depth of the main arm and, beside it, for the two-launch loop with a collective every step."""
    n = n_single // 8
    res = {"envs_per_gpu": n, "gpus": 8, "unit": "env-steps/s per GPU",
           "note": "1/8 of the batch on one GPU with the record gather queued (1-rank RCCL communicator)"}
    # (ring_of_8: the main arm with twice the ring -- a chunk's rasteriser launch is then as long as the whole batch's)
    arms = (("main", args.gather_every, args.loop != "pair"), ("policy_loop", args.gather_every, False), ("pair_gather_every_step", 1, False)) + \
           ((("ring_of_8", 8, True),) if args.loop != "pair" and args.gather_every != 8 and max(args.steps, 200) % 8 == 0 else ())
    for key, every, want_fused in arms:
        r = run_arm("probe", key, args, {"game": game, "C": C, "n": n, "every": every, "fused": want_fused}, hip)
        r["share_of_linear"] = r["value"] / (single_value if want_fused else single_pair_value)
        r["share_of"] = "value" if want_fused else "serialised (the two-launch loop on the whole batch)"
        res[key] = r
    res["value"] = res["main"]["value"]
    res["share_of_linear"] = res["main"]["share_of_linear"]
    res["share_of_linear_policy_loop"] = res["policy_loop"]["share_of_linear"]
    return res


ARMS = {"config": config_arm, "mixed": mixed_arm, "probe": probe_arm}


if __name__ == "__main__":
    sys.exit(main())
