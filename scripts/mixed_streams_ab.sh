# BASELINE config 5's per-GPU share (32 768 envs, K = 4 ring): three segments on three streams against all three on one, alternating, each a bench.py run of its own
mkdir -p gpurun_out/r06
for i in 1 2 3; do
  for m in 3 1; do
    timeout 300 python bench.py --no-cpu-baseline --game mixed --envs 32768 --with-gather --mixed-streams $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $m', round(d['value']/1e6,2), round(d['ms_per_step'],4), d.get('repeats',{}).get('ms_per_step_in_run_order'))"
  done
done | tee gpurun_out/r06/mixed_streams_ab.txt
