#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched game-step hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One "step" = one pass of the hot path over one batch: the step kernel (transition + auto-reset, actions generated on the
device by the counter-based rule of SURVEY 8d) followed by the frame rasteriser writing uint8[N,H,W,3] into HBM.
Workload: Breakout, 65 536 envs per GPU (weak scaling, the default: the batch shards embarrassingly, envs never interact)
or 65 536 envs in total (--scaling strong: SURVEY 8d's headline batch cut into N contiguous shards), env seeds 1234 + global
env index.  For N > 1 there is one process per GPU: started by any launcher that exports RANK / LOCAL_RANK / WORLD_SIZE
(torch.distributed.run does), or by this script itself when it finds no RANK in its environment.  The only exchange is the
per-step all-gather of the packed 8-byte {reward, done, lives} records -- tbx_gather, RCCL behind the C-ABI, no PyTorch --
queued between the step and the rasteriser so that it overlaps with the latter.

Protocol (mirroring the repeat-and-summarise shape of the reference's test/benchmark.py:119-148): an untimed
pre-roll of step-only frames so that the timed region sees mid-game states with episodes ending and auto-resets firing,
W warm-up steps, then R regions of exactly K steps, each bracketed by device-sync + rank barrier on both sides; a region's
time is the MAX over ranks; the reported value is the MEDIAN region (min / max alongside).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (the rasteriser) priced against HBM bandwidth with HIP events recorded on the launch
                  stream around every render launch of the timed regions,
  step_only    -- the same loop without the rasteriser (not bandwidth-bound: no roofline),
  cpu_baseline -- the CPU oracle (oracle/, a port: ctoybox itself cannot be built offline) on this box's host cores, a
                  bounded sample of the same 65 536-env workload, plus BASELINE config 1 (one env, one thread) -- N=1 only.
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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --envs per GPU; strong: --envs in total, sharded contiguously over the GPUs")
    ap.add_argument("--game", default="breakout")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--no-render", action="store_true", help="step-only mode as the main arm (no roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the step-only arm and the strong-scaling share probe")
    ap.add_argument("--with-gather", action="store_true", help="run the RCCL record gather even at one rank (1-rank communicator)")
    ap.add_argument("--protocol", default="batch", choices=["batch", "reference", "agent"],
                    help="'reference' = the raw single-env loop of the reference's harness (test/benchmark.py:44-58)")
    ap.add_argument("--deepmind", action="store_true",
                    help="agent protocol: also EpisodicLife + FireReset + NoopReset(30) + episode monitor (wrap_deepmind)")
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


def cpu_baseline(game, channels, n_total, target_seconds):
    """The CPU oracle on all usable host cores at the headline batch size: n_total envs held as chunks of 4 096 (one frame
    buffer of a chunk is reused, so host memory stays ~1.5 GB instead of n_total full frames), step + render with auto-reset
    and the same action rule, for about `target_seconds` of wall time (step count calibrated from a short probe)."""
    from toybox_amd import Engine
    lib = _oracle_lib()
    if lib is None:
        return None
    cores = usable_cores()
    os.environ["TBX_ORACLE_THREADS"] = str(cores)
    chunk = min(4096, n_total)
    n_chunks = max(1, n_total // chunk)
    engines = []
    for c in range(n_chunks):
        e = Engine(game, chunk, lib=lib)
        e.seed(SEED_BASE + c * chunk)
        e.new_game()
        engines.append(e)
    frame = np.empty((chunk, engines[0].height, engines[0].width, channels), np.uint8)

    def run(t_from, count):
        t0 = time.perf_counter()
        for t in range(t_from, t_from + count):
            for c, e in enumerate(engines):
                e.step_synthetic(ACTION_SEED, t, env_offset=c * chunk)
                e.render_device(frame.ctypes.data, channels)
        return time.perf_counter() - t0

    run(0, 1)                                           # warm-up (thread pool, page faults)
    probe = run(1, 2) / 2 + 1e-9                        # calibration
    steps = int(max(3, min(20000, target_seconds / probe)))
    dt = run(3, steps)
    for e in engines:
        e.close()
    n = chunk * n_chunks
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%s step+render(%dch), %d envs (as %d x %d) x %d steps, OpenMP static partition over envs, %.1f s" %
                      (game, channels, n, n_chunks, chunk, steps, dt)}


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

    def run(self, one_step, t0_index, K, R):
        times = []
        t = t0_index
        for _ in range(R):
            self.sync(); self.rank_barrier()
            w0 = time.perf_counter()
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

def bench_mixed(args, world, rank, local_rank):
    """BASELINE config 5: Breakout + Amidar + SpaceInvaders, args.envs envs per GPU split in three contiguous segments,
    three homogeneous launches per phase on three streams, one record gather per segment."""
    from toybox_amd import hip
    from toybox_amd.parallel import MixedBatch
    games = ["breakout", "amidar", "space_invaders"]
    per = args.envs // 3
    mb = MixedBatch(games, per, device=local_rank, global_offset=rank * per * 3)
    streams = [hip.Stream() for _ in games]
    mb.attach_streams([s.ptr for s in streams])
    gather = world > 1 or args.with_gather
    if gather:
        with quiet_stdout():
            mb.gather_init(rank, world)
    C, K, Wm, R = args.channels, args.steps, args.warmup, args.repeats
    render = not args.no_render
    lead = mb.engines[0]
    reg = Region(hip.synchronize, (lambda: lead.gather_reduce_max(0.0)) if gather else (lambda: None),
                 (lambda v: lead.gather_reduce_max(v)) if gather else (lambda v: v))

    def one(t):
        mb.step_synthetic(ACTION_SEED, t)
        if render:
            mb.render_device(C)

    for e, off in zip(mb.engines, mb.offsets):
        for t in range(args.preroll):
            e.step_synthetic(ACTION_SEED, t, env_offset=off, auto_reset=True)
    for t in range(Wm):
        one(args.preroll + t)
    times, _ = reg.run(one, args.preroll + Wm, K, R)
    mb.sync()
    if rank == 0:
        total = world * mb.n_envs
        fb = mb.frame_bytes(C) if render else 0
        rep = summarize(times, K)
        ms = rep["ms_per_step_median"]
        out = {"metric": "env steps/sec (whole node), mixed Breakout+Amidar+SpaceInvaders batch", "value": total / (ms * 1e-3),
               "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": ms, "repeats": rep,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+int32", "data": "synthetic",
               "config": {"workload": "mixed batch, %d envs/GPU = 3 x %d (breakout, amidar, space_invaders), %s, three streams%s"
                                      % (mb.n_envs, per, "step + RGB render" if render else "step-only",
                                         ", per-step RCCL gather of 8 B/env records" if gather else ""),
                          "envs_per_gpu": mb.n_envs, "envs_total": total},
               "roofline": ({"bound": "hbm", "kernel": "the three rasterisers together (whole-step time, not per kernel)",
                             "achieved": fb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None} if render else None)}
        print(json.dumps(out), flush=True)
    mb.close()
    return 0


def bench_reference_protocol(args):
    """Protocol A: test/benchmark.py:44-58 verbatim -- one env, action = legal[i % len(legal)],
    new_game() when game_over() else apply_ale_action(move), no rendering, FPS = steps / elapsed.  One FFI round trip
    per step, so on the GPU this measures launch + sync latency, not throughput; the CPU oracle runs the same loop."""
    from toybox_amd import Engine
    from toybox_amd import toybox as tbmod
    from toybox_amd.toybox import Toybox
    nsteps = max(args.steps, 1000)

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

    out = {"metric": "raw single-env steps/sec, reference harness protocol (test/benchmark.py:44-58)", "unit": "env-steps/s",
           "n_gpus": 1, "steps": nsteps, "warmup": 0, "higher_is_better": True, "vs_baseline": None, "data": "synthetic",
           "scaling": "weak", "config": {"workload": "%s single env, cycling legal actions, new_game on game over, no render" % args.game}}
    with Toybox(args.game) as tb:
        raw_loop(tb)
        out["value"] = raw_loop(tb)
    out["ms_per_step"] = 1000.0 / out["value"]
    lib = _oracle_lib()
    if lib is not None and not args.no_cpu_baseline:
        tbmod.set_engine_factory(lambda game, n: Engine(game, n, lib=lib))
        with Toybox(args.game) as tb:
            out["cpu_baseline"] = {"value": raw_loop(tb), "unit": "env-steps/s", "cores": 1, "kind": "port",
                                   "sample": "same loop over the CPU oracle, %d steps" % nsteps}
        tbmod.set_engine_factory(None)
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
                   noop_max=30 if dm else 0, noop_seed=2024)
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
           "config": {"workload": "%s fused %sMaxAndSkip(4)+WarpFrame(84)+ClipReward+FrameStack(4), %d envs, device actions"
                                  % (args.game, "NoopReset(30)+EpisodicLife+FireReset+Monitor+" if dm else "", n)}}
    print(json.dumps(out), flush=True)
    eng.close()
    return 0


# ---------------------------------------------------------------------------------------------- launcher

def spawn_ranks(args):
    """No RANK in the environment and --gpus N > 1: start the N ranks ourselves (before anything touches a GPU), pass rank
    0's line through, exit with the worst return code."""
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
    out, _ = procs[0].communicate()
    rc = max([p.wait() for p in procs])
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def main():
    args = parse()
    if args.protocol == "reference":
        return bench_reference_protocol(args)
    if args.protocol == "agent":
        return bench_agent_protocol(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)

    from toybox_amd import Engine, hip
    from toybox_amd.parallel import exchange_unique_id, forget_unique_id, shard_range, world_from_env
    rank, world, local_rank = world_from_env()
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    if os.environ.get("TBX_BENCH_ONE_DEVICE"):     # diagnostic: every rank on device 0 (exercises the N > 1 flow on a 1-GPU box)
        local_rank = 0
    hip.set_device(local_rank)

    game = args.game
    if game == "mixed":
        return bench_mixed(args, world, rank, local_rank)
    if args.scaling == "strong":
        start, end = shard_range(args.envs, world, rank)
        n_total = args.envs
        width = max(e - s for s, e in (shard_range(args.envs, world, r) for r in range(world)))
    else:
        start, end = rank * args.envs, (rank + 1) * args.envs
        n_total = world * args.envs
        width = args.envs
    n = end - start
    eng = Engine(game, n, device=local_rank)
    eng.seed(SEED_BASE + start)            # env i of this rank: seed 1234 + global index
    eng.new_game()
    H, W, C = eng.height, eng.width, args.channels
    render = not args.no_render
    gather = world > 1 or args.with_gather
    gather_note = None
    fw = None
    if gather:
        try:
            if os.environ.get("TBX_BENCH_NO_RCCL"):
                raise RuntimeError("disabled by TBX_BENCH_NO_RCCL")
            with quiet_stdout():
                uid = exchange_unique_id(rank, world, eng.gather_unique_id)
                eng.gather_init(world, rank, uid, records_per_rank=width)   # collective (ncclCommInitRank)
            forget_unique_id(rank)
        except Exception as ex:    # no usable RCCL: the shards still run; ranks start and stop together through files
            from toybox_amd.parallel import FileWorld
            gather_note = "RCCL communicator unavailable (%s): no per-step gather, file barrier between ranks" % (str(ex).splitlines()[0][:160],)
            print("bench.py: " + gather_note, file=sys.stderr)
            gather = False
            fw = FileWorld(rank, world)
    stream = hip.Stream()
    sp = stream.ptr
    K, Wm, R = args.steps, args.warmup, max(1, args.repeats)
    if gather:
        reg = Region(hip.synchronize, lambda: eng.gather_reduce_max(0.0), lambda v: eng.gather_reduce_max(v))
    elif fw is not None:
        reg = Region(hip.synchronize, fw.barrier, fw.allreduce_max)
    else:
        reg = Region(hip.synchronize, lambda: None, lambda v: v)

    # HIP events around every render launch of the timed regions, created up front (nothing is allocated inside a region)
    pool = [(hip.Event(), hip.Event()) for _ in range(K * R)] if render else []
    used = [0]

    def full_step(t):
        eng.step_synthetic(ACTION_SEED, t, env_offset=start, auto_reset=True, stream=sp)
        if gather:
            eng.gather(stream=sp)          # on the engine's communication stream: overlaps with the rasteriser below
        if render:
            i = used[0]
            if i < len(pool):
                pool[i][0].record(sp)
            eng.render_device(0, C, stream=sp)
            if i < len(pool):
                pool[i][1].record(sp)
                used[0] = i + 1

    def step_only(t):
        eng.step_synthetic(ACTION_SEED, t, env_offset=start, auto_reset=True, stream=sp)
        if gather:
            eng.gather(stream=sp)

    t = 0
    for _ in range(args.preroll):          # untimed: bring the batch to mid-game states (episodes end, auto-resets fire)
        eng.step_synthetic(ACTION_SEED, t, env_offset=start, auto_reset=True, stream=sp)
        t += 1
    used[0] = len(pool)                    # warm-up launches are not timed
    for _ in range(Wm):
        full_step(t)
        t += 1
    hip.synchronize()
    used[0] = 0
    times, t = reg.run(full_step, t, K, R)
    render_ms = float(np.mean([a.elapsed_ms(b) for a, b in pool[:used[0]]])) if render else None
    for a, b in pool:
        a.close(); b.close()
    rep = summarize(times, K)

    extras = {}
    if render and not args.no_extras:
        so_times, t = reg.run(step_only, t, K, R)
        so = summarize(so_times, K)
        extras["step_only"] = {"value": n_total / (so["ms_per_step_median"] * 1e-3), "unit": "env-steps/s",
                               "ms_per_step": so["ms_per_step_median"], "repeats": so,
                               "note": "same loop without the rasteriser; latency / issue bound, no roofline"}

    # sanity: the rollout really played (scores move, lives are lost, episodes end)
    eng.sync()
    score, lives, level, over = eng.scalars()
    check = {"mean_score": float(score.mean()), "mean_lives": float(lives.mean()), "max_level": int(level.max()),
             "frames_played": t}
    if t >= 300 and not (check["mean_score"] > 0):
        print("bench.py: the rollout did not play (mean score %.3f after %d frames)" % (check["mean_score"], t), file=sys.stderr)
        return 3
    frame_bytes = H * W * C if render else 0
    bytes_per_step = 2 * S_GAME[game] + A_BYTES + O_BYTES + frame_bytes
    eng.close()

    if rank == 0:
        ms = rep["ms_per_step_median"]
        out = {
            "metric": "env steps/sec (whole node), Breakout 64k-env batch" if game == "breakout" else "env steps/sec (whole node), %s" % game,
            "value": n_total / (ms * 1e-3),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
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
                            "%d-frame step-only pre-roll before the warm-up"
                            % (game, "step + %dx%dx%d uint8 frame render" % (H, W, C) if render else "step-only",
                               args.envs, "per GPU" if args.scaling == "weak" else "in total", args.preroll),
                "envs_per_gpu": n, "envs_total": n_total, "frame_hwc": [H, W, C] if render else None,
                "parallelism": ("env-sharded x%d, per-step RCCL all-gather of 8 B/env records behind the C-ABI (tbx_gather), "
                                "overlapped with the rasteriser" % world) if gather else
                               ("env-sharded x%d, no collective (%s)" % (world, gather_note)) if gather_note else "single GPU",
                "algorithmic_bytes_per_env_step": bytes_per_step,
            },
        }
        if render:
            achieved = n * frame_bytes / (render_ms * 1e-3) / 1e9    # GB/s, algorithmic frame bytes of one launch
            traffic, source = None, None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    rec = json.load(open(tp)).get("%s_render_%dch_%d" % (game, C, n))
                    if rec:
                        traffic = rec["hbm_bytes_per_launch"]
                        source = "profiles/traffic.json (static: rocprofv3 PMC pass %s, not measured in this run)" % rec.get("source", "")
                except Exception:
                    traffic = None
            out["roofline"] = {
                "bound": "hbm", "kernel": "%s render (%d ch)" % (game, C),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": source,
                "algorithmic_bytes_per_launch": n * frame_bytes, "avg_launch_ms": render_ms, "launches_timed": K * R,
            }
        else:
            out["roofline"] = None
        out.update(extras)
        out["check"] = check
        if world == 1 and not args.no_extras and args.scaling == "weak" and n >= 16384:
            try:
                out["strong_scaling_share"] = strong_share_probe(args, game, C, n)
            except Exception as ex:
                out["strong_scaling_share"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(game, C, n_total, args.cpu_seconds)
                out["cpu_config1"] = cpu_config1(game, C)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    return 0


def strong_share_probe(args, game, C, n_single):
    """What ONE GPU of an 8-GPU strong-scaling run of the same batch would do: n/8 envs with the per-step gather on
    (1-rank communicator: launch + stream-hop cost of the collective, no wire time).  8 x this value over the single-GPU value
    is the scaling efficiency the per-step fixed costs allow at that shard size."""
    from toybox_amd import Engine, hip
    n = n_single // 8
    eng = Engine(game, n, device=0)
    eng.seed(SEED_BASE)
    eng.new_game()
    with quiet_stdout():
        eng.gather_init(1, 0, eng.gather_unique_id())
    st = hip.Stream()

    def one(t):
        eng.step_synthetic(ACTION_SEED, t, auto_reset=True, stream=st.ptr)
        eng.gather(stream=st.ptr)
        eng.render_device(0, C, stream=st.ptr)

    for t in range(args.preroll):
        eng.step_synthetic(ACTION_SEED, t, auto_reset=True, stream=st.ptr)
    for t in range(20):
        one(args.preroll + t)
    reg = Region(hip.synchronize, lambda: eng.gather_reduce_max(0.0), lambda v: eng.gather_reduce_max(v))
    K = max(args.steps, 200)
    times, _ = reg.run(one, args.preroll + 20, K, 5)
    rep = summarize(times, K)
    eng.close()
    return {"envs": n, "value": n / (rep["ms_per_step_median"] * 1e-3), "unit": "env-steps/s per GPU",
            "ms_per_step": rep["ms_per_step_median"], "repeats": rep,
            "note": "1/8 of the batch on one GPU with the per-step record gather queued (1-rank RCCL communicator)"}


if __name__ == "__main__":
    sys.exit(main())
