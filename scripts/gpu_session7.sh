#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s7
for pl in 0 1 0 1; do
timeout 300 python bench.py --game mixed --envs 32766 --with-gather --no-cpu-baseline --pipeline $pl > gpurun_out/s7/mixed_$pl.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s7/mixed_$pl.json')); print('mixed pipeline $pl', round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['frac'],3), d['pipeline']['resolved_per_game'])"
done
for pl in 0 1; do
timeout 300 python bench.py --game mixed --envs 32766 --no-cpu-baseline --pipeline $pl > gpurun_out/s7/mixed_ng_$pl.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s7/mixed_ng_$pl.json')); print('mixed no-gather pipeline $pl', round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
done
