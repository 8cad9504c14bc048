#!/bin/bash
# HBM bytes per launch of the step and rasteriser kernels of one game: FETCH_SIZE / WRITE_SIZE in separate passes (they do not
# share a pass on gfx950), each under its own timeout.  usage: scripts/pmc_step_bytes.sh <game> [envs]
GAME=${1:-space_invaders}; ENVS=${2:-65536}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_bytes_$GAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT/$CTR" -- python3 $REPO/bench.py --game $GAME --envs $ENVS --no-cpu-baseline --no-extras --repeats 1 --steps 10 --warmup 2 > "$OUT/$CTR.log" 2>&1
done
python3 - "$OUT" "$ENVS" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = 'render' if 'render' in k else 'step' if '_step' in k else None
        if k: agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
n = int(sys.argv[2])
for k, v in sorted(agg.items()):
    kib = sum(v) / len(v)
    print(k[0], k[1], '%.5g KiB per launch' % kib, '= %.1f B per env (raw; FETCH_SIZE x2 on gfx950)' % (kib * 1024 / n))
PY
