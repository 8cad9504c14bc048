#!/usr/bin/env python3
"""Condense what scripts/measure_round.sh <tag> left under gpurun_out/ into profiles/:
  profiles/<tag>_agent_summary.md     kernel-trace --stats of the agent protocol, per game
  profiles/<tag>_pmc_sq_counters.txt  SQ instruction / wave counters of the step + render kernels (pmc_*_sq*.txt)
  profiles/<tag>_bench_lines.md       one table row per bench.py line of the call (all from the same box)
usage: collect_round.py r02"""
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
    for game in ("breakout", "space_invaders", "amidar", "gridworld"):
        stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_agent_%s" % (tag, game), "*", "*_kernel_stats.csv"))
        if not stats:
            continue
        out += ["## " + game, "", "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
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
            "| file | metric | value | ms/step (median) | min–max | roofline GB/s | frac | config |", "|---|---|---|---|---|---|---|---|"]
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
        rows.append("| %s | %s | %.4g %s | %.4f | %s | %s | %s | %s |" % (
            os.path.basename(f), j.get("metric"), j.get("value", 0), j.get("unit", ""), j.get("ms_per_step", 0),
            ("%.4f–%.4f" % (min(ms), max(ms))) if ms else "", ("%.0f" % rl["achieved"]) if rl.get("achieved") else "",
            ("%.3f" % rl["frac"]) if rl.get("frac") else "", json.dumps(j.get("config", {}))))
    return "\n".join(rows) + "\n"


def main():
    tag = sys.argv[1]
    dst = os.path.join(ROOT, "profiles")
    for name, text in (("agent_summary.md", agent_summary(tag)), ("pmc_sq_counters.txt", sq_counters(tag)),
                       ("bench_lines.md", bench_lines(tag))):
        with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as fh:
            fh.write(text + ("\n" if not text.endswith("\n") else ""))
        print("wrote", "%s_%s" % (tag, name))


if __name__ == "__main__":
    main()
