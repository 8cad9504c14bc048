#!/bin/bash
# Everything DESIGN.md section 6 quotes, measured in ONE gpurun call on ONE box (boxes differ by up to ~15 %, so numbers of
# different calls are not comparable): bench lines per game / batch size / protocol, loop-form sweeps, rocprofv3 kernel-trace +
# PMC passes, kernel-gap analysis, rasterisers against the previous round's build.
# usage (on the GPU box): bash scripts/measure_round.sh r04        then, back home: python scripts/collect_round.py r04
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline"
# ---- bench lines (the default line carries serialised, step_only, scaling_strong, cpu_baseline, cpu_config1)
python bench.py > "$OUT/bench_breakout_65536.json" 2> "$OUT/bench_breakout_65536.err"
for g in space_invaders amidar; do $B --game $g > "$OUT/bench_${g}_65536.json" 2>/dev/null; done
$B --game gridworld --no-extras > "$OUT/bench_gridworld_65536.json" 2>/dev/null
for g in breakout space_invaders amidar; do $B --game $g --envs 4096 --steps 400 > "$OUT/bench_${g}_4096.json" 2>/dev/null; done
python bench.py --game mixed --envs 32766 --with-gather --cpu-seconds 8 > "$OUT/bench_mixed_32766.json" 2>/dev/null
$B --game mixed --envs 32766 --with-gather --loop pair --gather-every 1 > "$OUT/bench_mixed_32766_pair_k1.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 > "$OUT/bench_breakout_8192_gather.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 --loop pair --gather-every 1 > "$OUT/bench_breakout_8192_gather_pair_k1.json" 2>/dev/null
for g in breakout space_invaders amidar gridworld; do
  python bench.py --protocol agent --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}.json" 2>/dev/null
  python bench.py --protocol agent --deepmind --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}_deepmind.json" 2>/dev/null
done
for g in breakout space_invaders amidar; do
  python bench.py --protocol reference --gym --game $g --reps 10 > "$OUT/reference_${g}.json" 2>/dev/null        # 10 reps x 10 000 steps, both arms, CPU beside
done
for g in breakout space_invaders amidar; do
  python bench.py --protocol host --game $g --envs 8192 --steps 20 --warmup 3 > "$OUT/host_${g}.json" 2>/dev/null        # PCIe-inclusive: frames / observations to host memory
done
# ---- loop forms and ring depths against each other, interleaved in one process per game
timeout 400 python scripts/strong_sweep.py breakout 4096 8192 16384 65536 2>&1 | grep '^{' > "$OUT/sweep_breakout.txt"
SS_RING=8 timeout 200 python scripts/strong_sweep.py breakout 8192 2>&1 | grep '^{' > "$OUT/sweep_breakout_ring8.txt"
timeout 400 python scripts/strong_sweep.py space_invaders 4096 8192 65536 2>&1 | grep '^{' > "$OUT/sweep_space_invaders.txt"
timeout 400 python scripts/strong_sweep.py amidar 4096 65536 2>&1 | grep '^{' > "$OUT/sweep_amidar.txt"
# ---- the rasterisers (and [step ; render]) against the previous round's build, interleaved
for g in breakout space_invaders amidar; do
  AB_PREROLL=400 timeout 300 python scripts/ab_render.py $g 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so > "$OUT/ab_render_$g.txt" 2>&1
  AB_STEP=1 AB_PREROLL=400 timeout 300 python scripts/ab_render.py $g 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so > "$OUT/ab_step_render_$g.txt" 2>&1
done
# ---- instruction issue rates of a compute unit (SALU / VALU alone and side by side), for the issue-time prices of DESIGN section 6
make -C scripts/ubench issue_rate > /dev/null 2>&1; timeout 120 scripts/ubench/issue_rate 2>&1 | grep -v "waves/SIMD 8" > "$OUT/issue_rate.txt"
# ---- what the loops' time is made of (kernel trace: durations in the loop, gaps)
bash scripts/gpu_gaps.sh > /dev/null 2>&1
cp "$REPO/gpurun_out/gaps/summary.txt" "$OUT/kernel_gaps.txt" 2>/dev/null
# ---- profiles: kernel trace + PMC (separate passes).  Default loop form (fused where the engine fuses) and the two-launch form.
bash scripts/profile_gpu.sh ${TAG} --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_pair --no-extras --loop pair --pipeline 0 > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_space_invaders --game space_invaders --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_amidar --game amidar --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_breakout_4096 --envs 4096 --no-extras > /dev/null 2>&1
for g in breakout space_invaders amidar; do bash scripts/profile_gpu.sh ${TAG}_${g}_4096_pair --game $g --envs 4096 --no-extras --loop pair --pipeline 0 > /dev/null 2>&1; done
bash scripts/profile_gpu.sh ${TAG}_breakout_8192_gather --envs 8192 --with-gather --no-extras > /dev/null 2>&1
# ---- SQ instruction counters of the rasterisers and of the agent observation kernels (issue roofline)
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES"; SQ2="SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; SQ3="SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
BENCH_ARGS="--no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_brk_sq "$SQ1" "$SQ2" > "$OUT/pmc_brk_sq.txt" 2>&1
BENCH_ARGS="--game space_invaders --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_si_sq "$SQ1" "$SQ2" "$SQ3" > "$OUT/pmc_si_sq.txt" 2>&1
BENCH_ARGS="--game amidar --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_ami_sq "$SQ1" "$SQ2" "$SQ3" > "$OUT/pmc_ami_sq.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for g in breakout space_invaders amidar gridworld; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_agent_$g" -- python3 $REPO/bench.py --protocol agent --deepmind --game $g --steps 60 --warmup 5 > /dev/null 2>&1
done
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_${TAG}_agent_si/$(echo $CTRS | cut -c1-12 | tr ' ' _)" -- python3 $REPO/bench.py --protocol agent --game space_invaders --steps 12 --warmup 3 > /dev/null 2>&1
done
python3 - "$REPO/gpurun_out/pmc_${TAG}_agent_si" > "$OUT/pmc_agent_si_sq.txt" <<'PY'
import csv, glob, sys, collections, re
agg, dur = collections.defaultdict(list), collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(?:_w\d)?(?:<[^>]*>)?)", r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(dur): print(k, 'avg_us %.1f' % (sum(dur[k]) / len(dur[k])), ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for (kk, c), v in sorted(agg.items()) if kk == k))
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_mixed" -- python3 $REPO/bench.py --game mixed --envs 32766 --with-gather --no-cpu-baseline --steps 50 --warmup 5 --repeats 2 > /dev/null 2>&1
cd "$REPO"
find gpurun_out -size +8M -delete; du -sh gpurun_out/$TAG gpurun_out/prof_${TAG}* 2>/dev/null | tail -20
