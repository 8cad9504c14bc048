#!/usr/bin/env python3
"""Condense what scripts/measure_round.sh <tag> left under gpurun_out/ into profiles/:
  profiles/<tag>_agent_summary.md     kernel-trace --stats of the agent protocol, per game
  profiles/<tag>_pmc_sq_counters.txt  SQ instruction / wave counters of the step + render kernels (pmc_*_sq*.txt)
  profiles/<tag>_bench_lines.md       one table row per bench.py line of the call (all from the same box), with the loop form
                                      `value` was measured with and the two-launch (`serialised`) rate beside it
  profiles/<tag>_loop_sweeps.txt      scripts/strong_sweep.py: loop forms x record-ring depths, interleaved per process
  profiles/<tag>_kernel_gaps.txt      kernel durations inside the loops and the gaps between them
  profiles/<tag>_ab_prev_round.txt    rasterisers and [step ; render] against the previous round's build
  profiles/<tag>*_summary.md, *_kernel_stats.csv, traffic.json   through scripts/summarize_profile.py, one per profile
usage: collect_round.py r04"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(\w+_kernel(?:_w\d)?(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def agent_summary(tag):
    out = ["# rocprofv3 kernel-trace --stats, agent protocol (`bench.py --protocol agent --deepmind --steps 60 --warmup 5`, 65 536 envs)", ""]
    for game, mode in [(g, m) for g in ("breakout", "space_invaders", "amidar", "gridworld") for m in ("", "ring_")]:
        stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_agent_%s%s" % (tag, mode, game), "*", "*_kernel_stats.csv"))
        if not stats:
            continue
        out += ["## " + game + (", the plane ring instead of the rolled stack (`--obs ring`; `*_agent_warp_kernel<0>`)" if mode else ""), "",
                "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(stats[0])):
            out.append("| `%s` | %s | %.1f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        out.append("")
    return "\n".join(out)


def sq_counters(tag):
    out = ["# SQ counters (scripts/pmc_gpu.sh: rocprofv3 --pmc with --kernel-trace only), average PER LAUNCH of the kernel class:",
           "# `render` = *_render_kernel, `step` = *_step*_kernel; divide by SQ_WAVES for per-wave figures", ""]
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "pmc_*sq*.txt"))):
        out += ["## " + os.path.basename(f)[:-4], open(f).read().rstrip(), ""]
    return "\n".join(out)


def bench_lines(tag):
    rows = ["# bench.py lines of `scripts/measure_round.sh %s` (one gpurun call, one box)" % tag, "",
            "`loop`: the form `value` was measured with (fused = tbx_render_step_synthetic, one launch per frame where the engine fuses; "
            "pair = tbx_step_synthetic ; tbx_render_device).  `serialised`: the pair form in stream order on the same engine -- what a "
            "loop whose actions depend on the frame gets.  `whole-step frac` = frame bytes of the batch / ms per step / 8 TB/s.", "",
            "| file | metric | value | ms/step (median) | min–max | loop | serialised value (ms/step) | rasteriser GB/s | kernel frac | whole-step frac | strong share | config |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "*.json"))):
        line = None
        for ln in open(f):
            ln = ln.strip()
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
        if not line:
            continue
        try:
            j = json.loads(line)
        except ValueError:
            continue
        rl = j.get("roofline") or {}
        rep = j.get("repeats") or {}
        ms = rep.get("ms_per_step_in_run_order") or []
        ser = j.get("serialised") or {}
        whole = None
        if rl.get("algorithmic_bytes_per_launch") and j.get("ms_per_step"):
            whole = rl["algorithmic_bytes_per_launch"] * (j.get("n_gpus", 1) if (j.get("config", {}).get("envs_per_gpu") == j.get("config", {}).get("envs_total")) else 1) / (j["ms_per_step"] * 1e-3) / 8e12
        elif rl.get("algorithmic_bytes_per_step") and j.get("ms_per_step"):
            whole = rl["algorithmic_bytes_per_step"] / (j["ms_per_step"] * 1e-3) / 8e12
        ss = (j.get("scaling_strong") or {}).get("share_of_linear")
        rows.append("| %s | %s | %.4g %s | %.4f | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            os.path.basename(f), j.get("metric"), j.get("value", 0), j.get("unit", ""), j.get("ms_per_step", 0),
            ("%.4f–%.4f" % (min(ms), max(ms))) if ms else "", (j.get("loop") or {}).get("form", ""),
            ("%.4g (%.4f)" % (ser["value"], ser["ms_per_step"])) if ser.get("value") else "",
            ("%.0f" % rl["achieved"]) if rl.get("achieved") else "", ("%.3f" % rl["frac"]) if rl.get("frac") else "",
            ("%.3f" % whole) if whole else "", ("%.3f" % ss) if ss else "", json.dumps(j.get("config", {}))))
    return "\n".join(rows) + "\n"


def cat_files(tag, pattern, header):
    out = [header, ""]
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, pattern))):
        out += ["## " + os.path.basename(f), open(f).read().rstrip(), ""]
    return "\n".join(out)


def loop_sweeps(tag):
    out = ["# scripts/strong_sweep.py (one process per game and call, rounds of [pair, fused] x [no gather, a collective per step, a K-step ring]): "
           "median ms per step and M env-steps/s", ""]
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "sweep_*.txt"))):
        out.append("## " + os.path.basename(f))
        for ln in open(f):
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            cells = ", ".join("%s %.4f (%.1f M)" % (k, v["median"], v["M_per_s"]) for k, v in d.items() if isinstance(v, dict))
            out.append("%s %d envs, %d steps: %s" % (d["game"], d["envs"], d["steps"], cells))
        out.append("")
    return "\n".join(out)


def host_rates(tag):
    out = ["# bench.py --protocol host (the PCIe link in the loop; never `value`): what VecEnv.step() hands a caller on the host, per arm", ""]
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "host_*.json"))):
        try:
            j = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
        except (IndexError, ValueError):
            continue
        out.append("## %s -- %s" % (os.path.basename(f), j.get("config", {}).get("workload", "")))
        for k, v in j.get("arms", {}).items():
            out.append("  %-44s %8.3f M %-14s %8.3f ms per step  %5.1f GB/s over the link" % (k, v["value"] / 1e6, v["unit"], v["ms_per_step"], v.get("pcie_GB_per_s", 0.0)))
        out.append("")
    return "\n".join(out)


def rehearsal(tag):
    out = ["# bench.py --gpus N --gather host --one-device: the whole N-process flow on ONE GPU (every rank on device 0, the record gather over the",
           "# host transport): what it exercises is the flow, not a scaling figure -- N ranks share one GPU.  Round 6: with --steps 20 every rank",
           "# runs its share as rollout chunks (a step lane and a rasteriser lane per process).  On ONE shared GPU that form degrades with the number",
           "# of processes -- 2 ranks 1.32 ms per step (49.5 M env-steps/s for the pair), 4 ranks 4.49 ms, 8 ranks 9.13 ms, against 3.53 ms for 8 ranks",
           "# of fused launches in stream order (--rollout-chunks off) -- presumably because every exchange of the host transport is a barrier among",
           "# the ranks' STEP launches, and on a shared GPU a process's step launch queues behind the other processes' rasteriser launches; with a GPU",
           "# per rank nothing of another rank runs in front of it.  Not a figure of merit either way: the exchange is verified in every run.", ""]
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "rehearsal_*.json"))):
        try:
            j = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
        except (IndexError, ValueError):
            out.append("## %s: no line" % os.path.basename(f))
            continue
        keep = {k: j.get(k) for k in ("n_gpus", "value", "ms_per_step", "scaling", "rccl", "gather", "share_of_linear", "loop")}
        keep["weak"] = {k: (j.get("weak") or {}).get(k) for k in ("value", "ms_per_step", "envs_total", "gather")}
        keep["config"] = {k: j["config"].get(k) for k in ("envs_per_gpu", "envs_total", "parallelism")}
        keep["cpu_baseline"] = (j.get("cpu_baseline") or {}).get("value")
        out += ["## " + os.path.basename(f), json.dumps(keep, indent=1), ""]
    return "\n".join(out)


PROFILES = [  # (profile tag suffix, traffic key, kernel substring)
    ("", "breakout_render_3ch_65536_fused", "render_step_kernel<3"), ("_pair", "breakout_render_3ch_65536", "render_kernel<3"),
    ("_space_invaders", "space_invaders_render_3ch_65536", "render_kernel<3"), ("_amidar", "amidar_render_3ch_65536", "render_kernel<3"),
    ("_breakout_4096", "breakout_render_3ch_4096_fused", "render_kernel<3"), ("_breakout_4096_pair", "breakout_render_3ch_4096", "render_kernel<3"),
    ("_space_invaders_4096_pair", "space_invaders_render_3ch_4096", "render_kernel<3"), ("_amidar_4096_pair", "amidar_render_3ch_4096", "render_kernel<3"),
    ("_breakout_8192_gather", "breakout_render_3ch_8192_fused", "render_step_kernel<3"), ("_mixed", None, "render"),
]


def main():
    tag = sys.argv[1]
    dst = os.path.join(ROOT, "profiles")
    import subprocess
    for suffix, key, sub in PROFILES:
        if os.path.isdir(os.path.join(ROOT, "gpurun_out", "prof_%s%s" % (tag, suffix))):
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), tag + suffix] + ([key, sub] if key else [])
            subprocess.run(cmd, stdout=subprocess.DEVNULL)
    for name, text in (("agent_summary.md", agent_summary(tag)), ("pmc_sq_counters.txt", sq_counters(tag)),
                       ("loop_sweeps.txt", loop_sweeps(tag)),
                       ("ab_prev_round.txt", cat_files(tag, "ab_*.txt", "# scripts/ab_render.py: lib_prev.so = the previous round's final build, interleaved with this build on one box")),
                       ("issue_rate.txt", cat_files(tag, "issue_rate.txt", "# scripts/ubench/issue_rate.hip: scalar-ALU and vector-ALU instructions a compute unit issues per nanosecond, alone and side by side")),
                       ("kernel_gaps.txt", cat_files(tag, "kernel_gaps.txt", "# scripts/gpu_gaps.sh + scripts/trace_gaps.py: kernel durations inside the loops and the idle gaps in front of them")),
                       ("host_rates.txt", host_rates(tag)), ("rehearsal.txt", rehearsal(tag)),
                       ("agent_diag.txt", cat_files(tag, "agent_diag_*.txt", "# scripts/agent_diag.sh: the fused agent observation kernel of SpaceInvaders with parts switched off (DIAG build), time per agent step and SQ counters per launch")),
                       ("write_align.txt", cat_files(tag, "write_align.txt", "# scripts/ubench/write_align.hip")),
                       ("rollout_forms.txt", cat_files(tag, "rollout_forms.txt", "# scripts/rollout_ab.py: the random-rollout loop of ONE engine in stream order / overlapped behind the device-side ticket / as rollout chunks of 4, interleaved, ms per step (first block: K = 4 record ring, second: no gather)")),
                       ("fused_lib_ab.txt", cat_files(tag, "fused_lib_ab.txt", "# scripts/fused_lib_ab.py: the fused loop of the previous round's build (A) and of this build (B), interleaved on one box")),
                       ("bench_lines.md", bench_lines(tag))):
        with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as fh:
            fh.write(text + ("\n" if not text.endswith("\n") else ""))
        print("wrote", "%s_%s" % (tag, name))


if __name__ == "__main__":
    main()
