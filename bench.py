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
order, the rate a policy-driven loop gets.  Whenever `value` is not the pair form in stream order, that is measured on the
same engine and reported beside it as `serialised`.

Protocol (mirroring the repeat-and-summarise shape of the reference's test/benchmark.py:119-148): an untimed
pre-roll of step-only frames so that the timed region sees mid-game states with episodes ending and auto-resets firing,
W warm-up steps, then R regions of exactly K steps, each bracketed by device-sync + rank barrier on both sides; a region's
time is the MAX over ranks; the reported value is the MEDIAN region (min / max alongside).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       -- the dominant kernel (the rasteriser; fused: rasteriser + step launch) priced against HBM bandwidth with HIP
                    events recorded on the caller's stream around every 8th such launch of the timed regions,
  loop           -- which loop form `value` was measured with,
  serialised     -- the two-launch loop in stream order on the same engine (when `value` is anything else),
  step_only      -- the same loop without the rasteriser (not bandwidth-bound: no roofline),
  weak / strong  -- (N > 1) the other reading, and share_of_linear,
  scaling_strong -- (N = 1) what ONE GPU does with 1/8 of the batch plus the record gather, as a share of linear (a fraction),
  rccl           -- ranks the communicator spans as RCCL reports it, ring depth, bytes per collective, library, verified,
  cpu_baseline   -- the CPU oracle (oracle/, a port: ctoybox itself cannot be built offline) on this box's host cores, a
                    bounded sample of the same 65 536-env workload, plus BASELINE config 1 (one env, one thread) -- N=1 only.
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
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="which reading is `value`.  strong (default): --envs IN TOTAL, sharded contiguously over the GPUs -- the metric "
                         "names one 64k-env batch on 1/2/4/8 GPUs; weak: --envs per GPU.  With N > 1 the other one is measured too")
    ap.add_argument("--loop", default="auto", choices=["auto", "fused", "pair"],
                    help="auto / fused: tbx_render_step_synthetic where the engine fuses (Breakout RGB / RGBA); pair: step ; render")
    ap.add_argument("--gather-every", type=int, default=4,
                    help="K of the record ring (TBX_OPT_GATHER_EVERY): one RCCL all-gather per K steps (1 = every step)")
    ap.add_argument("--game", default="breakout")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--no-render", action="store_true", help="step-only mode as the main arm (no roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the step-only arm and the strong-scaling share probe")
    ap.add_argument("--with-gather", action="store_true", help="run the RCCL record gather even at one rank (1-rank communicator)")
    ap.add_argument("--allow-no-gather", action="store_true",
                    help="N > 1 only: if no RCCL communicator can be made, run without the per-step gather (file barrier) instead of failing")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1, 2, 3], help="TBX_OPT_PIPELINE of the main arm (1 = engine's choice)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / barrier walk-through without any GPU call")
    ap.add_argument("--gym", action="store_true", help="reference protocol: also the env.step() arm (test/benchmark.py:83-97)")
    ap.add_argument("--reps", type=int, default=30, help="reference protocol: repetitions (mean and s.e.m. reported)")
    ap.add_argument("--protocol", default="batch", choices=["batch", "reference", "agent", "host"],
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
        chunk = min(4096, n_seg)
        for c in range(max(1, n_seg // chunk)):
            e = Engine(game, chunk, lib=lib)
            e.seed(SEED_BASE + off + c * chunk)
            e.new_game()
            engines.append((e, off + c * chunk))
            total += chunk
            if game not in frames:
                frames[game] = np.empty((chunk, e.height, e.width, channels), np.uint8)

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
            mb.gather_init(rank, world, gather_every=args.gather_every)
    C, K, Wm, R = args.channels, args.steps, args.warmup, args.repeats
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
               "pipeline": {"option": args.pipeline, "resolved_per_game": dict(zip(games, pipe))},
               "loop": {"form": "fused" if fused else "pair", "fused_per_game": dict(zip(games, [e.get_option(_abi_mod().OPT_RENDER_STEP_FUSED) for e in mb.engines]))},
               "config": {"workload": "mixed batch, %d envs/GPU = 3 x %d (breakout, amidar, space_invaders), %s, three streams%s"
                                      % (mb.n_envs, per, "step + RGB render" if render else "step-only",
                                         (", RCCL gather of 8 B/env records, one collective per %d steps" % max(1, args.gather_every)) if gather else ""),
                          "envs_per_gpu": mb.n_envs, "envs_total": total},
               "roofline": ({"bound": "hbm", "kernel": "the three rasterisers together (whole-step time, not per kernel)",
                             "achieved": fb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "algorithmic_bytes_per_step": fb,
                             "per_kernel": "profiles/ (rocprofv3 kernel trace of this command: the three rasterisers' own durations and "
                                           "whether they overlap)"} if render else None),
               "rccl": ({"nranks": lead.gather_nranks(), "gather_bytes_per_step": 8 * per * 3 * world, "lib": lead.gather_library(),
                         "communicators": len(games)} if gather else None)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline([(g, per, i * per) for i, g in enumerate(games)], C, args.cpu_seconds)
            except Exception as ex:
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    mb.close()
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


def bench_host_protocol(args):
    """The PCIe-inclusive rate: what a caller on the HOST side of the boundary gets, the way the reference's consumers sit
    (ToyboxBaseEnv / VecEnv hand every frame to numpy: envs/atari/base.py:109, vec_env/__init__.py:63-74).  One step = actions
    from a host array, the batch step, every env's frame copied to host memory -- through ToyboxVecEnv.step().  Two arms: a
    fresh pageable array per step (the VecEnv contract as the reference's learners use it) and one reused page-locked array
    (reuse_obs_buffer=True); and the agent pipeline, whose observation is 7 KB per env instead of 100-200 KB.  Never `value`
    of the headline: that one is measured with everything resident in HBM."""
    import numpy as np
    from toybox_amd.envs import ToyboxPreprocVecEnv, ToyboxVecEnv
    n, K, Wm = args.envs, args.steps, args.warmup
    rng = np.random.default_rng(ACTION_SEED)
    arms = {}
    for name, reuse in (("pageable_fresh_array", False), ("pinned_reused_array", True)):
        env = ToyboxVecEnv(args.game, n, grayscale=False, seed=SEED_BASE, reuse_obs_buffer=reuse)
        na = env.action_space.n
        obs = env.reset()
        acts = [rng.integers(0, na, n) for _ in range(8)]
        for t in range(Wm):
            obs, _, _, _ = env.step(acts[t % 8])
        t0 = time.perf_counter()
        for t in range(K):
            obs, _, _, _ = env.step(acts[t % 8])
        dt = time.perf_counter() - t0
        arms[name] = {"value": n * K / dt, "unit": "env-steps/s", "ms_per_step": 1000 * dt / K, "host_GB_per_s": obs.nbytes * K / dt / 1e9}
        frame_bytes = obs.nbytes // n
        env.close()
    for name, reuse in (("agent_obs_84x84x4_pageable", False), ("agent_obs_84x84x4_pinned", True)):
        env = ToyboxPreprocVecEnv(args.game, n, seed=SEED_BASE, reuse_obs_buffer=reuse)
        na = env.action_space.n
        obs = env.reset()
        acts = [rng.integers(0, na, n) for _ in range(8)]
        for t in range(Wm):
            obs, _, _, _ = env.step(acts[t % 8])
        t0 = time.perf_counter()
        for t in range(K):
            obs, _, _, _ = env.step(acts[t % 8])
        dt = time.perf_counter() - t0
        arms[name] = {"value": n * K / dt, "unit": "agent-steps/s", "ms_per_step": 1000 * dt / K, "host_GB_per_s": obs.nbytes * K / dt / 1e9}
        env.close()
    best = arms["pinned_reused_array"]
    out = {"metric": "env steps/sec INCLUDING the PCIe transfer of every frame to the host, %s" % args.game, "value": best["value"],
           "unit": "env-steps/s", "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": best["ms_per_step"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic", "arms": arms,
           "config": {"workload": "%s ToyboxVecEnv.step(host actions) -> RGB frames in host memory, %d envs, %d B per frame"
                                  % (args.game, n, frame_bytes)}}
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
    HIP events bracket every EVERY-th rasteriser (fused: rasteriser + step) launch."""

    EVERY = 8       # one launch in EVERY carries the two HIP events (a pair costs the stream 5-10 us: around every launch
                    # that was 1 % of the 65 536-env step and 5 % of the 8 192-env one)

    def __init__(self, eng, hip, stream, start, channels, gather, render, n_steps, fused=False):
        self.eng, self.sp, self.start, self.C, self.gather, self.render = eng, stream.ptr, start, channels, gather, render
        self.fused = bool(fused and render)
        self.pool = [(hip.Event(), hip.Event()) for _ in range((n_steps + self.EVERY - 1) // self.EVERY)] if render else []
        self.used = len(self.pool)              # nothing is timed until arm() is called
        self.calls = 0

    def arm(self):
        self.used = 0
        self.calls = 0

    def full_step(self, t):
        e, sp = self.eng, self.sp
        if not self.fused:
            e.step_synthetic(ACTION_SEED, t, env_offset=self.start, auto_reset=True, stream=sp)
            if self.gather:
                e.gather(stream=sp)            # on the engine's communication stream: overlaps with the rasteriser below
        if self.render:
            i = self.used
            timed = i < len(self.pool) and self.calls % self.EVERY == 0
            self.calls += 1
            if timed:
                self.pool[i][0].record(sp)
            if self.fused:
                e.render_step_synthetic(ACTION_SEED, t, channels=self.C, env_offset=self.start, auto_reset=True, stream=sp)
            else:
                e.render_device(0, self.C, stream=sp)
            if timed:
                self.pool[i][1].record(sp)
                self.used = i + 1
        if self.fused and self.gather:
            e.gather(stream=sp)                # the records of the step that rode in the launch above

    def step_only(self, t):
        self.eng.step_synthetic(ACTION_SEED, t, env_offset=self.start, auto_reset=True, stream=self.sp)
        if self.gather:
            self.eng.gather(stream=self.sp)

    def render_ms(self):
        return float(np.mean([a.elapsed_ms(b) for a, b in self.pool[:self.used]])) if self.render and self.used else None

    def close(self):
        for a, b in self.pool:
            a.close(); b.close()
        self.pool = []


def timed_arm(eng, hip, reg, stream, start, C, gather, render, pipeline, t, K, Wm, R, fused=False):
    """Warm-up + R regions of K steps with TBX_OPT_PIPELINE = pipeline (pair form) or the fused call.  Returns (summary, avg
    rasteriser launch ms, resolved pipeline mode, next t)."""
    from toybox_amd import _abi
    eng.set_option(_abi.OPT_PIPELINE, 0 if fused else pipeline)
    mode = eng.get_option(_abi.OPT_PIPELINE_ACTIVE)
    loop = Loop(eng, hip, stream, start, C, gather, render, K * R, fused=fused)
    for _ in range(Wm):
        loop.full_step(t)
        t += 1
    hip.synchronize()
    loop.arm()
    times, t = reg.run(loop.full_step, t, K, R)
    rms = loop.render_ms()
    timed_arm.launches_timed = loop.used
    loop.close()
    return summarize(times, K), rms, mode, t


PIPELINE_NOTE = {0: "off: every call in stream order", 2: "the step runs beside the previous frame's rasteriser (internal step stream, "
                 "two sets of render records and step outputs)", 3: "the step runs beside the previous frame's rasteriser and consecutive "
                 "rasteriser launches alternate between two internal streams and two frame buffers"}
FUSED_NOTE = ("tbx_render_step_synthetic: the rasteriser of frame t and the batch step to frame t+1 are ONE launch (the step's blocks "
              "in front of the rasteriser's, other records buffer); random-action rollouts only -- `serialised` is the two-launch loop")


def make_communicator(eng, rank, world, width, gather_every, tag):
    """tbx_gather_init over `world` ranks (id from rank 0 through the rendezvous file), K-step record ring if asked.  Returns the
    `rccl` object of the JSON line; raises when no communicator over `world` ranks comes out of it."""
    from toybox_amd import _abi
    from toybox_amd.parallel import exchange_unique_id, forget_unique_id
    if os.environ.get("TBX_BENCH_NO_RCCL"):
        raise RuntimeError("disabled by TBX_BENCH_NO_RCCL")
    eng.set_option(_abi.OPT_GATHER_EVERY, max(1, gather_every))
    with quiet_stdout():
        uid = exchange_unique_id(rank, world, eng.gather_unique_id, tag=tag)
        eng.gather_init(world, rank, uid, records_per_rank=width)   # collective (ncclCommInitRank)
    forget_unique_id(rank, tag=tag)
    K = eng.gather_every()
    rccl = {"nranks": eng.gather_nranks(), "records_per_rank": width, "gather_every": K,
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


def run_reading(args, hip, game, rank, world, local_rank, scaling, tag, with_extras):
    """One reading of the metric on this rank: engine for its shard, communicator (N > 1 or --with-gather) with one verified
    exchange, pre-roll, the timed arm (fused where the engine fuses, unless --loop pair), and with_extras the serialised
    two-launch loop and the step-only loop beside it.  Returns a dict (rank 0 assembles the line) or an int return code."""
    from toybox_amd import Engine, _abi
    from toybox_amd.parallel import FileWorld, shard_range
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
    H, W, C = eng.height, eng.width, args.channels
    render = not args.no_render
    gather = world > 1 or args.with_gather
    gather_note, rccl, fw = None, None, None
    if gather:
        try:
            rccl = make_communicator(eng, rank, world, width, args.gather_every, tag)
            rccl["verified"] = verify_gather(eng, hip, rank, world, n, sizes, start)
            eng.new_game()                 # (the verification stepped K frames)
        except Exception as ex:
            msg = str(ex).splitlines()[0][:200] if str(ex) else repr(ex)
            if world > 1 and not args.allow_no_gather:
                # the north star's 8-GPU number INCLUDES the collective: a run that cannot make it is a failed run
                print("bench.py: rank %d: no verified RCCL gather over %d ranks (%s); pass --allow-no-gather to measure the shards "
                      "without the per-step gather" % (rank, world, msg), file=sys.stderr)
                return 4
            gather_note = "RCCL communicator unavailable (%s): no per-step gather, file barrier between ranks" % msg
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
    rep, render_ms, mode, t = timed_arm(eng, hip, reg, stream, start, C, gather, render, args.pipeline, t, K, Wm, R, fused=fused)
    res = {"n": n, "n_total": n_total, "start": start, "H": H, "W": W, "C": C, "render": render, "gather": gather, "rccl": rccl,
           "gather_note": gather_note, "rep": rep, "render_ms": render_ms, "mode": mode, "fused": fused,
           "launches_timed": timed_arm.launches_timed, "extras": {}}
    frame_bytes = H * W * C if render else 0
    if with_extras and (fused or mode != 0):
        # the same engine, two launches per frame in stream order: what a policy-driven loop (actions computed from the frame) gets
        srep, s_ms, _, t = timed_arm(eng, hip, reg, stream, start, C, gather, render, 0, t, K, Wm, R, fused=False)
        sms = srep["ms_per_step_median"]
        res["extras"]["serialised"] = {"value": n_total / (sms * 1e-3), "unit": "env-steps/s", "ms_per_step": sms, "repeats": srep,
                                       "avg_launch_ms": s_ms,
                                       "roofline_frac": (n * frame_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if s_ms else None,
                                       "note": "tbx_step_synthetic ; tbx_render_device in stream order (TBX_OPT_PIPELINE = 0): the rate of a loop whose "
                                               "actions depend on the frame"}
        eng.set_option(_abi.OPT_PIPELINE, args.pipeline)
    if render and with_extras:
        loop = Loop(eng, hip, stream, start, C, gather, False, 0)
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
    args = parse()
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
    if os.environ.get("TBX_BENCH_ONE_DEVICE"):     # diagnostic: every rank on device 0 (exercises the N > 1 flow on a 1-GPU box)
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
                            "%d-frame step-only pre-roll before the warm-up"
                            % (game, "step + %dx%dx%d uint8 frame render" % (H, W, C) if render else "step-only",
                               args.envs, "per GPU" if args.scaling == "weak" else "in total", args.preroll),
                "envs_per_gpu": n, "envs_total": n_total, "frame_hwc": [H, W, C] if render else None,
                "parallelism": ("env-sharded x%d, RCCL all-gather of 8 B/env records behind the C-ABI (tbx_gather), %s, overlapped with the "
                                "rasteriser" % (world, "one collective per step" if K_ring == 1 else "K-step record ring: one collective per %d steps" % K_ring))
                               if gather else ("env-sharded x%d, no collective (%s)" % (world, r["gather_note"])) if r["gather_note"] else "single GPU",
                "algorithmic_bytes_per_env_step": bytes_per_step,
            },
            "loop": {"form": "fused" if fused else "pair", "what": FUSED_NOTE if fused else "tbx_step_synthetic ; tbx_render_device, two launches per frame"},
            "pipeline": {"option": args.pipeline, "resolved": mode, "what": PIPELINE_NOTE.get(mode),
                         "applies_to": "the two-launch loop form only; see `serialised`"},
            "rccl": r["rccl"],
        }
        if render:
            render_ms = r["render_ms"]
            achieved = n * frame_bytes / (render_ms * 1e-3) / 1e9    # GB/s, algorithmic frame bytes of one launch
            traffic, source = None, None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    rec = json.load(open(tp)).get("%s_render_%dch_%d%s" % (game, C, n, "_fused" if fused else ""))
                    if rec:
                        traffic = rec["hbm_bytes_per_launch"]
                        source = "profiles/traffic.json (static: rocprofv3 PMC pass %s, not measured in this run)" % rec.get("source", "")
                except Exception:
                    traffic = None
            out["roofline"] = {
                "bound": "hbm", "kernel": "%s render (%d ch)%s" % (game, C, " + step, one launch" if fused else ""),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": source,
                "algorithmic_bytes_per_launch": n * frame_bytes, "avg_launch_ms": render_ms, "launches_timed": r["launches_timed"],
                "timing": "HIP events on the caller's stream around every %dth rasteriser launch of the timed regions" % Loop.EVERY +
                          ("" if mode != 3 else "; launches overlap in this mode, so this is the time from one launch's end to the "
                                               "next one's end (what a launch costs in steady state), not a kernel's own duration"),
            }
        else:
            out["roofline"] = None
        out.update(r["extras"])
        out["check"] = r["check"]
        if other is not None:
            oms = other["rep"]["ms_per_step_median"]
            key = "weak" if args.scaling == "strong" else "strong"
            out[key] = {"value": other["n_total"] / (oms * 1e-3), "unit": "env-steps/s", "ms_per_step": oms, "repeats": other["rep"],
                        "envs_per_gpu": other["n"], "envs_total": other["n_total"], "rccl": other["rccl"],
                        "loop": "fused" if other["fused"] else "pair",
                        "note": "the other reading of the metric, measured in the same invocation with its own communicator"}
            s_, w_ = (out, out[key]) if args.scaling == "strong" else (out[key], out)
            # strong total over weak total = what N GPUs make of ONE batch against N x a full batch each: the share of linear scaling
            out["share_of_linear"] = s_["value"] / w_["value"]
        if world == 1 and not args.no_extras and n >= 16384 and ENGINE_FACTORY is None:
            try:
                out["scaling_strong"] = strong_share_probe(args, game, C, n, out["value"])
            except Exception as ex:
                out["scaling_strong"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline and ENGINE_FACTORY is None:
            try:
                out["cpu_baseline"] = cpu_baseline([(game, n_total, 0)], C, args.cpu_seconds)
                out["cpu_config1"] = cpu_config1(game, C)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    return 0


def strong_share_probe(args, game, C, n_single, single_value):
    """What ONE GPU of an 8-GPU run of the SAME batch would do: n/8 envs with the record gather on (1-rank communicator: launch
    and stream-hop cost of the collective, no wire time).  share_of_linear = that rate over the single-GPU rate of the whole
    batch (a fraction: 1.0 = eight GPUs are eight times one).  Measured for the loop form and ring depth of the main arm and,
    beside it, for the two-launch loop with a collective every step."""
    from toybox_amd import Engine, _abi, hip
    n = n_single // 8
    res = {"envs_per_gpu": n, "gpus": 8, "unit": "env-steps/s per GPU",
           "note": "1/8 of the batch on one GPU with the record gather queued (1-rank RCCL communicator)"}
    K = max(args.steps, 200)
    arms = (("main", args.gather_every, args.loop != "pair"), ("pair_gather_every_step", 1, False))
    for key, every, want_fused in arms:
        eng = Engine(game, n, device=0)
        eng.seed(SEED_BASE)
        eng.new_game()
        eng.set_option(_abi.OPT_GATHER_EVERY, max(1, every))
        with quiet_stdout():
            eng.gather_init(1, 0, eng.gather_unique_id())
        st = hip.Stream()
        for t in range(args.preroll):
            eng.step_synthetic(ACTION_SEED, t, auto_reset=True, stream=st.ptr)
        reg = Region(hip.synchronize, lambda: eng.gather_reduce_max(0.0), lambda v: eng.gather_reduce_max(v))
        fused = want_fused and C >= 3 and eng.get_option(_abi.OPT_RENDER_STEP_FUSED) == 1
        rep, rms, mode, _ = timed_arm(eng, hip, reg, st, 0, C, True, True, args.pipeline, args.preroll, K, 20, 5, fused=fused)
        v = n / (rep["ms_per_step_median"] * 1e-3)
        res[key] = {"value": v, "ms_per_step": rep["ms_per_step_median"], "repeats": rep, "loop": "fused" if fused else "pair",
                    "gather_every": eng.gather_every(), "pipeline_resolved": mode, "avg_launch_ms": rms, "share_of_linear": v / single_value}
        eng.close()
    res["value"] = res["main"]["value"]
    res["share_of_linear"] = res["main"]["share_of_linear"]
    return res


if __name__ == "__main__":
    sys.exit(main())
