#!/bin/bash
# SQ instruction counters of the fused agent observation kernel per DIAG variant (scripts/agent_diag.py) -- on the GPU box.
GAME=${1:-space_invaders}; N=${2:-65536}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/agent_diag; mkdir -p $OUT
cd $REPO && python scripts/agent_diag.py $GAME $N > $OUT/times_$GAME.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for B in 0 1 2 4 8 16; do
  AGENT_DIAG_ONE=$B rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_${GAME}_$B/a -- python3 $REPO/scripts/agent_diag.py $GAME $N > /dev/null 2>&1
  AGENT_DIAG_ONE=$B rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_${GAME}_$B/b -- python3 $REPO/scripts/agent_diag.py $GAME $N > /dev/null 2>&1
done
python3 - $OUT $GAME > $OUT/counters_$GAME.txt <<'PY'
import csv, glob, sys, collections
out, game = sys.argv[1], sys.argv[2]
for b in (0, 1, 2, 4, 8, 16):
    agg, dur = collections.defaultdict(list), []
    for f in glob.glob('%s/pmc_%s_%d/*/*/*counter_collection.csv' % (out, game, b)):
        for r in csv.DictReader(open(f)):
            if 'agent_warp_kernel' not in r['Kernel_Name']: continue
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    if dur:
        print('diag %2d  kernel avg %.1f us  ' % (b, sum(dur) / len(dur)) + '  '.join('%s=%.4g' % (k, sum(v) / len(v)) for k, v in sorted(agg.items())))
PY
find $OUT -size +4M -delete
