#!/bin/bash
# usage: scripts/pmc_gpu.sh <tag> "<counters pass1>" ["<counters pass2>" ...]  -- runs bench.py (10 steps) under rocprofv3 --pmc
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 $BENCH_ARGS > "$OUT/p$i.log" 2>&1
  python3 - "$OUT/p$i" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        k='render' if 'render' in k else 'step' if '_step' in k else None
        if k: agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k[0],k[1],'%.4g'%(sum(v)/len(v)))
PY
done
