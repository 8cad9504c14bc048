#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched game-step hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: the step kernel (transition + auto-reset,
actions generated on the device by the counter-based rule of SURVEY 8d) followed by the frame
rasteriser writing uint8[N,H,W,3] into HBM.  Workload at every N: Breakout, 65536 envs PER GPU
(weak scaling: the batch shards embarrassingly, envs never interact), env seeds 1234 + global env
index.  For N > 1 the driver launches one process per GPU through torch.distributed.run; the only
exchange is the per-step all_gather of the packed {reward, done, lives} record over RCCL.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (the rasteriser) priced against HBM bandwidth with HIP events
                  recorded around every render launch of the timed region, and
  cpu_baseline -- the CPU oracle (oracle/, a port: ctoybox itself cannot be built offline) timed on
                  this box's host cores on a bounded sample of the same workload (N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

# algorithmic bytes per env-step: 2*S_game + A + O + F
S_GAME = {"breakout": 72, "space_invaders": 248, "amidar": 420, "gridworld": 17}   # gridworld: player 8 + score 4 + over 4 + one cell
A_BYTES, O_BYTES = 1, 5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--game", default="breakout")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--no-render", action="store_true", help="step-only mode (reported separately, no roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--protocol", default="batch", choices=["batch", "reference", "agent"],
                    help="'reference' = the raw single-env loop of the reference's harness (test/benchmark.py:44-58)")
    ap.add_argument("--deepmind", action="store_true",
                    help="agent protocol: also EpisodicLife + FireReset + NoopReset(30) + episode monitor (wrap_deepmind)")
    ap.add_argument("--force-dist", action="store_true", help="run the torch.distributed/RCCL gather path even at world size 1")
    ap.add_argument("--cpu-envs", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=40)
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


def cpu_baseline(game, channels, n_envs, steps, target_seconds=12.0):
    """Times the CPU oracle (step + render, auto-reset, same action rule) on all usable host cores for about
    `target_seconds` of wall time (a bounded sample: the step count is calibrated from a short probe)."""
    from toybox_amd import Engine, _abi
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        return None
    cores = usable_cores()
    os.environ["TBX_ORACLE_THREADS"] = str(cores)
    lib = ctypes.CDLL(path)
    _abi.bind(lib)
    e = Engine(game, n_envs, lib=lib)
    e.seed(1234)
    e.new_game()
    def run(t_from, count):
        t0 = time.perf_counter()
        for t in range(t_from, t_from + count):
            e.step_synthetic(1337, t)
            e.render_device(0, channels)
        return time.perf_counter() - t0

    run(0, 4)                                           # warm-up (thread pool, page faults)
    probe = run(4, 8) + 1e-9                            # calibration
    steps = int(max(steps, min(20000, target_seconds / max(probe / 8, 1e-6))))
    dt = run(12, steps)
    e.close()
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%s step+render(%dch), %d envs x %d steps, OpenMP static partition over envs, %.1f s" %
                      (game, channels, n_envs, steps, dt)}


def bench_mixed(args, world, rank, local_rank, use_dist, dist):
    """BASELINE config 5: Breakout + Amidar + SpaceInvaders, args.envs envs per GPU split in three contiguous segments,
    three homogeneous launches per phase on three streams."""
    from toybox_amd import hip
    from toybox_amd.parallel import MixedBatch
    games = ["breakout", "amidar", "space_invaders"]
    per = args.envs // 3
    mb = MixedBatch(games, per, device=local_rank, global_offset=rank * per * 3)
    streams = [hip.Stream() for _ in games]
    mb.attach_streams([s.ptr for s in streams])
    C, K, Wm = args.channels, args.steps, args.warmup
    render = not args.no_render

    def barrier():
        if use_dist:
            dist.barrier()
        hip.synchronize()

    def one(t):
        mb.step_synthetic(1337, t)
        if render:
            mb.render_device(C)

    for t in range(Wm):
        one(t)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        one(Wm + i)
    barrier()
    elapsed = time.perf_counter() - t0
    mb.sync()
    if use_dist:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        total = world * mb.n_envs
        fb = mb.frame_bytes(C) if render else 0
        out = {"metric": "env steps/sec (whole node), mixed Breakout+Amidar+SpaceInvaders batch", "value": total * K / elapsed,
               "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1000.0 * elapsed / K,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+int32", "data": "synthetic",
               "config": {"workload": "mixed batch, %d envs/GPU = 3 x %d (breakout, amidar, space_invaders), %s, three streams"
                                      % (mb.n_envs, per, "step + RGB render" if render else "step-only"),
                          "envs_per_gpu": mb.n_envs, "envs_total": total},
               "roofline": ({"bound": "hbm", "kernel": "the three rasterisers together (whole-step time, not per kernel)",
                             "achieved": fb / (elapsed / K) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fb / (elapsed / K) / 1e9 / HBM_PEAK_GBS, "traffic": None} if render else None)}
        print(json.dumps(out), flush=True)
    mb.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def bench_reference_protocol(args):
    """Protocol A: test/benchmark.py:44-58 verbatim -- one env, action = legal[i % len(legal)],
    new_game() when game_over() else apply_ale_action(move), no rendering, FPS = steps / elapsed.  One FFI round trip
    per step, so on the GPU this measures launch + sync latency, not throughput; the CPU oracle runs the same loop."""
    from toybox_amd import Engine, _abi
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
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if os.path.exists(path) and not args.no_cpu_baseline:
        lib = ctypes.CDLL(path)
        _abi.bind(lib)
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
    eng.seed(1234)
    dm = bool(args.deepmind)
    eng.agent_init(skip=4, out_h=84, out_w=84, stack=4, clip_reward=True, episodic_life=dm, fire_reset=dm,
                   noop_max=30 if dm else 0, noop_seed=2024)
    eng.agent_reset()
    stream = hip.Stream()
    for t in range(Wm):
        eng.agent_step_synthetic(1337, t, stream=stream.ptr)
    hip.synchronize()
    t0 = time.perf_counter()
    for t in range(Wm, Wm + K):
        eng.agent_step_synthetic(1337, t, stream=stream.ptr)
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


def main():
    args = parse()
    if args.protocol == "reference":
        return bench_reference_protocol(args)
    if args.protocol == "agent":
        return bench_agent_protocol(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, args.gpus),
              file=sys.stderr)
        return 2

    from toybox_amd import Engine, _abi, hip
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)                       # --force-dist without a launcher: a world of one
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    hip.set_device(local_rank)

    n = args.envs
    game = args.game
    if game == "mixed":
        return bench_mixed(args, world, rank, local_rank, use_dist, dist)
    eng = Engine(game, n, device=local_rank)
    eng.seed(1234 + rank * n)            # env i of this rank: seed 1234 + global index
    eng.new_game()
    env_offset = rank * n
    H, W, C = eng.height, eng.width, args.channels
    render = not args.no_render

    if use_dist:
        stream_ptr = torch.cuda.current_stream().cuda_stream
        packed_local = torch.zeros(n, dtype=torch.int64, device="cuda")
        gathered = torch.zeros(n * world, dtype=torch.int64, device="cuda")
        packed_src, _ = eng.device_buffer(_abi.BUF_PACKED)
    else:
        stream = hip.Stream()
        stream_ptr = stream.ptr

    K, Wm = args.steps, args.warmup
    ev = [(hip.Event(), hip.Event()) for _ in range(K)] if render else []
    pending = None

    def one_step(t, events=None):
        nonlocal pending
        eng.step_synthetic(1337, t, env_offset=env_offset, auto_reset=True, stream=stream_ptr)
        if use_dist:
            if pending is not None:
                pending.wait()
            hip.memcpy_dtod_async(packed_local.data_ptr(), packed_src, 8 * n, stream_ptr)
            pending = dist.all_gather_into_tensor(gathered, packed_local, async_op=True)
        if render:
            if events:
                events[0].record(stream_ptr)
            eng.render_device(0, C, stream=stream_ptr)
            if events:
                events[1].record(stream_ptr)

    def barrier():
        if use_dist:
            dist.barrier()
        hip.synchronize()

    for t in range(Wm):
        one_step(t)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        one_step(Wm + i, ev[i] if render else None)
    if pending is not None:
        pending.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.sync()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    render_ms = None
    if render:
        render_ms = float(np.mean([a.elapsed_ms(b) for a, b in ev]))

    # sanity: the rollout really played (scores move, episodes end)
    score, lives, level, over = eng.scalars()
    frame_bytes = H * W * C if render else 0
    bytes_per_step = 2 * S_GAME[game] + A_BYTES + O_BYTES + frame_bytes

    if rank == 0:
        value = world * n * K / elapsed
        out = {
            "metric": "env steps/sec (whole node), Breakout 64k-env batch" if game == "breakout" else "env steps/sec (whole node), %s" % game,
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": 1000.0 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if game == "breakout" else "int32",
            "data": "synthetic",
            "config": {
                "workload": "%s %s, %d envs/GPU, uniform random legal actions generated on device "
                            "(splitmix64 counter rule, seed 1337), env seeds 1234+global index, auto-reset on done"
                            % (game, "step + %dx%dx%d uint8 frame render" % (H, W, C) if render else "step-only", n),
                "envs_per_gpu": n, "envs_total": n * world, "frame_hwc": [H, W, C] if render else None,
                "parallelism": "env-sharded x%d, per-step RCCL all_gather of 8 B/env records" % world if world > 1 else "single GPU",
                "algorithmic_bytes_per_env_step": bytes_per_step,
            },
        }
        if render:
            achieved = n * frame_bytes / (render_ms * 1e-3) / 1e9    # GB/s, algorithmic frame bytes of one launch
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    rec = json.load(open(tp)).get("%s_render_%dch_%d" % (game, C, n))
                    traffic = rec["hbm_bytes_per_launch"] if rec else None
                except Exception:
                    traffic = None
            out["roofline"] = {
                "bound": "hbm", "kernel": "%s render (%d ch)" % (game, C),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": n * frame_bytes, "avg_launch_ms": render_ms,
            }
        else:
            out["roofline"] = None
        out["check"] = {"mean_score": float(score.mean()), "mean_lives": float(lives.mean()), "max_level": int(level.max())}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(game, C, args.cpu_envs, args.cpu_steps)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)

    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
