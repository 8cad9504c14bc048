#!/usr/bin/env python3
"""Static instruction counts of one kernel, attributed to the source functions they were inlined from (by .loc line tables).
usage: asm_attrib.py file.hip kernel_mangled_substring   (compiles with -gline-tables-only -S for gfx950)
Straight-line code (set-up, loads, stores) executes once per wave, so its static count is its dynamic count."""
import collections, os, re, subprocess, sys
src_path, kern = sys.argv[1], sys.argv[2]
out = "/tmp/_attrib.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-w",
                       "-gline-tables-only", "--offload-device-only", "-S", "-o", out, os.path.abspath(src_path)] + sys.argv[3:], cwd=os.path.dirname(os.path.abspath(src_path)) or ".")
lines = open(out).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
st = next(i for i, l in enumerate(lines) if kern in l and l.startswith("_Z") and l.rstrip().split(";")[0].rstrip().endswith(":"))
en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
def functions(path):
    fs = []
    for i, l in enumerate(open(path).read().split("\n"), 1):
        m = re.match(r"\s*(?:static\s+)?(?:__device__|__global__|__host__).*?\b(\w+)\s*\(", l)
        if m: fs.append((i, m.group(1)))
    return fs
here = os.path.dirname(os.path.abspath(src_path))
fcache = {}
def fn(fid, line):
    name = files.get(fid, "?")
    if name not in fcache:
        p = os.path.join(here, name)
        fcache[name] = functions(p) if os.path.exists(p) else []
    best = name
    for a, n in fcache[name]:
        if a <= line: best = n
        else: break
    return best
cur = (0, 0)
v, s, o = collections.Counter(), collections.Counter(), collections.Counter()
for l in lines[st:en]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    t = l.strip()
    if not t or t[0] in ".;/" or t.split(";")[0].rstrip().endswith(":"): continue
    op = t.split()[0]
    k = fn(*cur)
    if op.startswith("v_"): v[k] += 1
    elif op.startswith("s_"): s[k] += 1
    else: o[k] += 1
print("static: VALU %d  SALU %d  other (memory, LDS) %d" % (sum(v.values()), sum(s.values()), sum(o.values())))
for k in sorted(set(v) | set(s) | set(o), key=lambda k: -(v[k] + s[k])):
    print("%-32s VALU %5d  SALU %5d  other %4d" % (k, v[k], s[k], o[k]))
