#!/bin/bash
# usage: scripts/pmc_agent.sh <game> -- SQ instruction-mix counters of the agent-protocol kernels (rocprofv3 --pmc passes)
GAME=${1:-space_invaders}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_agent_$GAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 $REPO/bench.py --protocol agent --game $GAME --steps 6 --warmup 2 > "$OUT/p$i.log" 2>&1
  python3 - "$OUT/p$i" <<'PY'
import csv,glob,sys,collections,re
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel)',r['Kernel_Name']); k=m.group(1) if m else r['Kernel_Name'][:30]
        if 'warp' in k or 'step' in k or 'render' in k: agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k[0],k[1],'%.4g'%(sum(v)/len(v)))
PY
done
