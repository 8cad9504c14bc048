#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s8
for form in 0 2; do
PS_STEP_FORM=$form PS_MODES=0,2 PS_ROUNDS=3 timeout 300 python scripts/pipeline_sweep.py amidar 16384 65536 2>&1 | grep '^{' >> gpurun_out/s8/sweep_amidar2.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/s8/sweep_amidar2.txt'):
    d=json.loads(ln)
    print(d['game'], d['envs'], 'form', d['step_form'], ' '.join('m%s=%.4f(%.3f)'%(k[4:],v['median_ms'],v['frac_of_8TBs']) for k,v in d.items() if k.startswith('mode')))
PY
