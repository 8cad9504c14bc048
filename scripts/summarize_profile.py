#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_gpu.sh) into profiles/<tag>_*.{csv,md}
and update profiles/traffic.json (HBM bytes per launch of the dominant kernel from the PMC passes).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB;
FETCH_SIZE under-reports wide coalesced streaming reads by 2x (doubled here, flagged as an upper
bound for non-streaming reads); WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(\w+_kernel(?:_w\d)?(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def main():
    tag = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else None        # e.g. breakout_render_3ch_65536
    kernel_sub = sys.argv[3] if len(sys.argv) > 3 else "render_kernel<3"
    src = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = ["# rocprofv3 summary `%s`" % tag, ""]
    host = os.path.join(src, "host.txt")
    if os.path.exists(host):
        lines += ["host: " + open(host).read().strip(), ""]
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
        lines += ["## kernel-trace --stats (%s)" % (open(os.path.join(src, "cmd.txt")).read().strip() if os.path.exists(os.path.join(src, "cmd.txt")) else "bench.py --steps 100 --warmup 10"), "",
                  "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(stats[0])):
            lines.append("| `%s` | %s | %.1f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                             float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        lines.append("")
    # a big Breakout RGB render goes out as TWO launches of the same kernel (1 024 envs, then the rest): the table above
    # averages over both, so the rasteriser launches are listed once more by grid size, with the sum of a render's parts
    traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if traces:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            k = short(r["Kernel_Name"])
            if "render_kernel" in k:
                by[(k, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        names = collections.defaultdict(list)
        for (k, g), v in by.items():
            names[k].append((g, v))
        multi = {k: parts for k, parts in names.items() if len(parts) > 1}
        if multi:
            lines += ["## rasteriser launches by grid size (kernel trace)", "", "| kernel | grid (threads) | calls | avg us |", "|---|---|---|---|"]
            for k, parts in sorted(multi.items()):
                total = 0.0
                for g, v in sorted(parts):
                    lines.append("| `%s` | %d | %d | %.1f |" % (k, g, len(v), sum(v) / len(v)))
                    total += sum(v) / len(v)
                lines.append("| `%s` | one render = the sum of its parts | | **%.1f** |" % (k, total))
            lines.append("")
    # the dominant kernel inside the TIMED regions against what bench.py's own line of the same run says (roofline.avg_launch_ms):
    # --stats above averages over every launch of the process, the settle and warm-up launches included
    log = os.path.join(src, "trace.log")
    line = None
    if os.path.exists(log):
        for ln in open(log, errors="replace"):
            ln = ln.strip()
            if ln.startswith("{") and '"metric"' in ln:
                try:
                    line = json.loads(ln)
                except ValueError:
                    pass
    if traces and line and line.get("roofline"):
        rl = line["roofline"]
        K, R = int(line.get("steps", 0)), int((line.get("repeats") or {}).get("n", 0))
        fused = (line.get("loop") or {}).get("form") == "fused"
        want = "render_step_kernel" if fused else "render_kernel"
        rows = [r for r in csv.DictReader(open(traces[0])) if want in short(r["Kernel_Name"]) and ("render_step" in short(r["Kernel_Name"])) == fused]
        # one render = all launches of the rasteriser between two step kernels (big Breakout renders go out in two parts): group by
        # launch order into renders of `parts` launches each
        grids = sorted({int(r["Grid_Size_X"]) for r in rows})
        parts = len(grids) if not fused else 1
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        renders = [sum(durs[i:i + parts]) for i in range(0, len(durs) - parts + 1, parts)]
        timed = renders[-K * R:] if K * R and len(renders) >= K * R else renders
        if timed:
            avg_t, avg_all = sum(timed) / len(timed), sum(renders) / len(renders)
            bytes_l = rl.get("algorithmic_bytes_per_launch", 0)
            lines += ["## the dominant kernel in the timed regions vs bench.py's own line of this run", "",
                      "| | launches | avg us | achieved GB/s | frac of 8 TB/s |", "|---|---|---|---|---|",
                      "| kernel trace, the %d timed launches (%d regions x %d steps) | %d | %.1f | %.0f | **%.4f** |" % (len(timed), R, K, len(timed), avg_t, bytes_l / avg_t / 1e3, bytes_l / avg_t / 1e3 / 8000.0),
                      "| kernel trace, every launch of the process (settle + warm-up included) | %d | %.1f | %.0f | %.4f |" % (len(renders), avg_all, bytes_l / avg_all / 1e3, bytes_l / avg_all / 1e3 / 8000.0),
                      "| bench.py `roofline` (HIP events, %s launches in %s spans) | %s | %.1f | %.0f | **%.4f** |" % (rl.get("launches_timed"), rl.get("event_spans"), rl.get("launches_timed"), 1e3 * rl["avg_launch_ms"], rl["achieved"], rl["frac"]),
                      "| bench.py `ms_per_step` (median region, host clock) | | %.1f | | |" % (1e3 * line["ms_per_step"]), ""]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "render_kernel" in k:
                k = "%s @ grid %s" % (k, r["Grid_Size"])       # (the parts of a two-part launch are different rows)
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
    traffic = {}
    if per:
        lines += ["## PMC passes (separate runs, --pmc with --kernel-trace only)", "",
                  "| kernel | grid | wg | VGPR | SGPR | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (fetch x2 corrected) | avg us (profiled) |",
                  "|---|---|---|---|---|---|---|---|---|"]
        for k, c in sorted(per.items()):
            if "kernel" not in k:
                continue
            mean = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else None
            fs, ws = mean("FETCH_SIZE"), mean("WRITE_SIZE")
            hbm = None
            if fs is not None and ws is not None:
                hbm = (2.0 * fs + ws) * 1024.0
            v, s, l, g, w = meta[k]
            lines.append("| `%s` | %s | %s | %s | %s | %s | %s | %s | %.1f |" % (
                k, g, w, v, s, "%.0f" % fs if fs is not None else "-", "%.0f" % ws if ws is not None else "-",
                "%.4g" % hbm if hbm else "-", mean("_dur_us")))
            if kernel_sub in re.sub(r"_w\d<", "<", k) and hbm:       # (brk_render_kernel_w5<3,..> is a render_kernel<3)
                # a render = all the launches of the rasteriser it takes (two for big Breakout RGB batches): bytes add up
                traffic = {"hbm_bytes_per_launch": hbm + traffic.get("hbm_bytes_per_launch", 0.0),
                           "fetch_kib_raw": fs + traffic.get("fetch_kib_raw", 0.0), "write_kib_raw": ws + traffic.get("write_kib_raw", 0.0),
                           "correction": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE exact; summed over the launches of one render",
                           "source": "profiles/%s_summary.md" % tag,
                           # what the figure is tied to: the kernel sources of the profiled tree and the kernel's register counts
                           "csrc_sha16": (open(os.path.join(src, "csrc_sha16.txt")).read().strip() or None) if os.path.exists(os.path.join(src, "csrc_sha16.txt")) else None,
                           "kernel": k, "vgpr": v, "sgpr": s}
            extra = {n: mean(n) for n in c if n not in ("FETCH_SIZE", "WRITE_SIZE", "_dur_us")}
            if extra:
                lines.append("|  | | | | | " + ", ".join("%s=%.4g" % kv for kv in sorted(extra.items())) + " | | | |")
        lines.append("")
    open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    if key and traffic:
        tp = os.path.join(dst, "traffic.json")
        db = json.load(open(tp)) if os.path.exists(tp) else {}
        db[key] = traffic
        json.dump(db, open(tp, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
