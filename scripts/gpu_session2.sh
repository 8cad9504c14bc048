#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s2; rm -f gpurun_out/s2/prio.txt
for g in "" 1; do
  PS_GATHER=$g PS_ROUNDS=4 timeout 300 python scripts/pipeline_sweep.py breakout 4096 8192 16384 65536 2>&1 | grep '^{' >> gpurun_out/s2/prio.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/s2/prio.txt'):
    d=json.loads(ln)
    print(d['lib'][-14:], d['envs'], 'gather' if d['gather'] else '      ', ' '.join('m%s=%.4f(%.3f)'%(k[4:],v['median_ms'],v['frac_of_8TBs']) for k,v in d.items() if k.startswith('mode')))
PY
timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_sharding.py -m gpu -x -q 2>&1 | tail -3
