#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s5
for g in breakout space_invaders amidar; do
  timeout 300 python bench.py --protocol reference --gym --game $g --reps 5 --steps 3000 > gpurun_out/s5/ref_$g.json 2> gpurun_out/s5/ref_$g.err
  python -c "
import json; d=json.load(open('gpurun_out/s5/ref_$g.json')); print('$g raw %.0f +- %.0f  gym %.0f +- %.0f  cpu raw %.0f gym %.0f' % (d['value'], d['sem'], d['gym']['value'], d['gym']['sem'], d['cpu_baseline']['value'], d['cpu_baseline']['gym']['value']))"
done
timeout 300 python bench.py --game space_invaders --no-cpu-baseline --no-extras > gpurun_out/s5/bench_si.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s5/bench_si.json')); print('si 65536', d['value'], d['ms_per_step'], d['pipeline']['resolved'], d['roofline']['frac'])"
timeout 300 python bench.py --game space_invaders --envs 4096 --no-cpu-baseline --no-extras > gpurun_out/s5/bench_si4096.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s5/bench_si4096.json')); print('si 4096', d['value'], d['ms_per_step'], d['pipeline']['resolved'], d['roofline']['frac'])"
timeout 300 python bench.py --game mixed --envs 32766 --with-gather --no-cpu-baseline > gpurun_out/s5/bench_mixed.json 2> gpurun_out/s5/bench_mixed.err; python -c "
import json; d=json.load(open('gpurun_out/s5/bench_mixed.json')); print('mixed', d['value'], d['ms_per_step'], d['roofline']['frac'], d['rccl'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s5/prof_mixed -- python3 $GRAFT_REPO_ROOT/bench.py --game mixed --envs 32766 --with-gather --no-cpu-baseline --steps 50 --warmup 5 --repeats 2 > $GRAFT_REPO_ROOT/gpurun_out/s5/prof_mixed.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/s5 -name "*kernel_stats.csv" | head; f=$(find gpurun_out/s5/prof_mixed -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
find gpurun_out/s5 -size +6M -delete
