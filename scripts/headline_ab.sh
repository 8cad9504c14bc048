# the headline loop (65 536 envs, no gather) and 32 768 envs: the fused launch in stream order against rollout chunks (one rasteriser launch per frame
# behind the previous one, the step launch beside them), alternating, each a bench.py run of its own
mkdir -p gpurun_out/r06
for n in 65536 32768; do
for i in 1 2; do
  for m in off on; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --envs $n --rollout-chunks $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$n chunks $m', round(d['value']/1e6,3), round(d['ms_per_step'],4), round(r['frac'],4), r.get('kernel'), d.get('config',{}).get('loop'), d.get('repeats',{}).get('ms_per_step_in_run_order'))"
  done
done
done | tee gpurun_out/r06/headline_ab.txt
