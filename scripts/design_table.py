#!/usr/bin/env python3
"""The table of DESIGN.md section 6 from the bench lines of one measurement pass (gpurun_out/<tag>/*.json).  usage: design_table.py r05"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def L(name):
    f = os.path.join(ROOT, "gpurun_out", tag, name)
    if not os.path.exists(f):
        return None
    ls = [ln for ln in open(f) if ln.startswith("{")]
    return json.loads(ls[-1]) if ls else None


def M(v):
    return "%.1f M" % (v / 1e6)


rows = ["| workload | env-steps/s | ms/step | kernel frac of 8 TB/s (events) / whole step | serialised (two launches) |", "|---|---|---|---|---|"]
h = L("bench_breakout_65536.json")
whole = lambda j: (j["roofline"].get("algorithmic_bytes_per_launch") or j["roofline"].get("algorithmic_bytes_per_step")) / (j["ms_per_step"] * 1e-3) / 8e12
if h:
    s = h["serialised"]
    rows.append("| **Breakout 65 536** (headline line, fused; it also carries the config rows) | **%s** | %.4f | **%.3f** / %.3f | %s (%.4f; kernel %.3f) |" % (
        M(h["value"]), h["ms_per_step"], h["roofline"]["frac"], whole(h), M(s["value"]), s["ms_per_step"], s["roofline_frac"]))
for name, label in (("bench_space_invaders_65536.json", "SpaceInvaders 65 536"), ("bench_amidar_65536.json", "Amidar 65 536"), ("bench_gridworld_65536.json", "GridWorld 65 536")):
    j = L(name)
    if j:
        rows.append("| %s | %s | %.4f | %.3f / %.3f | = value |" % (label, M(j["value"]), j["ms_per_step"], j["roofline"]["frac"], whole(j)))
if h and "configs" in h:
    for key, label in (("2_breakout_4096", "Breakout 4 096 (config 2)"), ("3_space_invaders_4096", "SpaceInvaders 4 096 (config 3)"),
                       ("4_amidar_4096", "Amidar 4 096 (config 4)"), ("5_mixed_32768_per_gpu", "mixed 32 768 + gather (config 5 per GPU, K = 4)")):
        c = h["configs"].get(key)
        if not c or "error" in c:
            continue
        ser = c.get("serialised")
        rows.append("| %s | %s | %.4f | %s / %.3f | %s |" % (
            label, M(c["value"]), c["ms_per_step"], ("%.3f" % c["kernel_frac"]) if c.get("kernel_frac") else "—", c["whole_step_frac"],
            ("%s (%.4f)" % (M(ser["value"]), ser["ms_per_step"])) if isinstance(ser, dict) else "= value"))
g = L("bench_breakout_8192_gather.json")
if g:
    rows.append("| Breakout 8 192 + gather (1/8 batch; fused, K = 4) | %s per GPU | %.4f | %.3f / %.3f | pair K = 4: %.4f; pair, a collective per step: %.4f |" % (
        M(g["value"]), g["ms_per_step"], g["roofline"]["frac"], whole(g), (L("bench_breakout_8192_gather_pair_k4.json") or {}).get("ms_per_step", 0),
        (L("bench_breakout_8192_gather_pair_k1.json") or {}).get("ms_per_step", 0)))
print("\n".join(rows))
if h:
    ss = h["scaling_strong"]
    print("\n**Strong-scaling share** (one GPU doing 1/8 of the batch with the gather on; in the headline line): fused + K = 4 **%.3f** of `value`; "
          "the policy loop (two launches + K = 4) **%.3f** of `serialised`; two launches with a collective per step %.3f." % (
              ss["main"]["share_of_linear"], ss["policy_loop"]["share_of_linear"], ss["pair_gather_every_step"]["share_of_linear"]))
    cb, c1 = h["cpu_baseline"], h["cpu_config1"]
    print("**CPU beside it** (`cpu_baseline`, the scalar C oracle under OpenMP on %d cores, same workload): %.2f M env-steps/s; BASELINE config 1 "
          "(one env, one thread): %.0f k step-only, %.0f k with the frame." % (cb["cores"], cb["value"] / 1e6, c1["step_only"] / 1e3, c1["step_render"] / 1e3))
ag = []
for g_ in ("breakout", "space_invaders", "amidar", "gridworld"):
    a, d = L("agent_%s.json" % g_), L("agent_%s_deepmind.json" % g_)
    if a and d:
        ag.append("%s %.1f / %.1f M" % (g_, a["value"] / 1e6, d["value"] / 1e6))
print("**Agent path** (65 536 envs, skip 4, 84×84×4; plain / every wrapper, agent-steps/s): " + ", ".join(ag) + ".")
rg = []
for g_ in ("breakout", "space_invaders", "amidar", "gridworld"):
    a, d = L("agent_%s_ring.json" % g_), L("agent_%s_ring_deepmind.json" % g_)
    if a and d:
        rg.append("%s %.1f / %.1f M" % (g_, a["value"] / 1e6, d["value"] / 1e6))
if rg:
    print("**Agent path, ring of planes instead of the rolled stack** (`--obs ring`, `new_plane = 2`): " + ", ".join(rg) + ".")
ref = []
for g_ in ("breakout", "space_invaders", "amidar"):
    r = L("reference_%s.json" % g_)
    if r:
        ref.append("%s raw %.0f k, `env.step()` %.1f k (CPU oracle %.0f k)" % (g_, r["value"] / 1e3, (r.get("gym") or {}).get("value", 0) / 1e3, (r.get("cpu_baseline") or {}).get("value", 0) / 1e3))
print("**One env** (`bench.py --protocol reference --gym`, 10 × 10 000 steps/s): " + "; ".join(ref) + ".")
