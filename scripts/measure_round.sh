#!/bin/bash
# Everything DESIGN.md section 6 quotes, measured in ONE gpurun call on ONE box (boxes differ by up to ~15 %, so numbers of
# different calls are not comparable): bench lines per game / batch size / protocol, then rocprofv3 kernel-trace + PMC passes.
# usage (on the GPU box): bash scripts/measure_round.sh r03
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline"
python bench.py > "$OUT/bench_breakout_65536.json" 2> "$OUT/bench_breakout_65536.err"
for g in space_invaders amidar; do $B --game $g > "$OUT/bench_${g}_65536.json" 2>/dev/null; done
$B --game gridworld --no-extras > "$OUT/bench_gridworld_65536.json" 2>/dev/null
for g in breakout space_invaders amidar; do $B --game $g --envs 4096 --no-extras > "$OUT/bench_${g}_4096.json" 2>/dev/null; done
python bench.py --game mixed --envs 32766 --with-gather --cpu-seconds 8 > "$OUT/bench_mixed_32766.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras > "$OUT/bench_breakout_8192_gather.json" 2>/dev/null
$B --with-gather --no-extras > "$OUT/bench_breakout_65536_gather.json" 2>/dev/null
for g in breakout space_invaders amidar gridworld; do
  python bench.py --protocol agent --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}.json" 2>/dev/null
  python bench.py --protocol agent --deepmind --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}_deepmind.json" 2>/dev/null
done
for g in breakout space_invaders amidar; do
  python bench.py --protocol reference --gym --game $g > "$OUT/reference_${g}.json" 2>/dev/null        # 30 reps x 10 000 steps, both arms, CPU beside
done
python bench.py --protocol reference --game gridworld --reps 10 > "$OUT/reference_gridworld.json" 2>/dev/null
# pipelined mode against stream order, interleaved in one process per game
timeout 300 python scripts/pipeline_sweep.py breakout 4096 8192 16384 32768 65536 2>&1 | grep '^{' > "$OUT/pipeline_breakout.txt"
PS_GATHER=1 timeout 300 python scripts/pipeline_sweep.py breakout 8192 65536 2>&1 | grep '^{' > "$OUT/pipeline_breakout_gather.txt"
PS_MODES=0,2 timeout 300 python scripts/pipeline_sweep.py space_invaders 4096 16384 65536 2>&1 | grep '^{' > "$OUT/pipeline_space_invaders.txt"
for form in 1 2; do PS_STEP_FORM=$form PS_MODES=0 PS_ROUNDS=3 timeout 300 python scripts/pipeline_sweep.py amidar 8192 16384 24576 32768 49152 65536 2>&1 | grep '^{' >> "$OUT/amidar_step_forms.txt"; done
python scripts/render_probe.py space_invaders 3 65536 400 3 5 7 9 12 18 > "$OUT/split_space_invaders.txt" 2>&1
python scripts/render_probe.py space_invaders 3 4096 400 3 5 7 9 12 18 >> "$OUT/split_space_invaders.txt" 2>&1
# the rasterisers against the previous round's build, interleaved
for g in breakout space_invaders amidar; do AB_PREROLL=400 timeout 300 python scripts/ab_render.py $g 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so > "$OUT/ab_render_$g.txt" 2>&1; done
AB_ENVS=4096 AB_PREROLL=400 timeout 300 python scripts/ab_render.py space_invaders 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so > "$OUT/ab_render_space_invaders_4096.txt" 2>&1
# the rasteriser's rate by output buffer, [step ; render] and render only, this build and the build before the first-wave stagger
# (scripts/ab/r03b/libtoybox_amd.so, commit 4c779ae), and the step;render / render-only / what-sits-between-two-launches table
make -C scripts/ubench rate_addr rate_state > /dev/null 2>&1
( cd scripts/ubench
  for g in 0 2 1 3; do for st in 1 0; do
    echo -n "game $g step=$st this build:      "; timeout 200 ./rate_addr 65536 $g $st 0 | grep -E "round 1" | awk '{printf "%s ", $(NF-3)}'; echo
    echo -n "game $g step=$st before stagger:  "; LD_LIBRARY_PATH=$PWD/../ab/r03b timeout 200 ./rate_addr 65536 $g $st 0 | grep -E "round 1" | awk '{printf "%s ", $(NF-3)}'; echo
  done; done
  for n in 8192 16384; do for st in 1 0; do
    echo -n "breakout $n step=$st this build:      "; timeout 200 ./rate_addr $n 0 $st 0 | grep -E "round 1" | awk '{printf "%s ", $(NF-3)}'; echo
    echo -n "breakout $n step=$st before stagger:  "; LD_LIBRARY_PATH=$PWD/../ab/r03b timeout 200 ./rate_addr $n 0 $st 0 | grep -E "round 1" | awk '{printf "%s ", $(NF-3)}'; echo
  done; done ) > "$OUT/rate_addr.txt" 2>&1
( cd scripts/ubench; timeout 200 ./rate_state 65536 | grep "round 2"; echo "-- before stagger"; LD_LIBRARY_PATH=$PWD/../ab/r03b timeout 200 ./rate_state 65536 | grep "round 2" ) > "$OUT/rate_state.txt" 2>&1
BENCH_ARGS="--game amidar --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_ami_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > "$OUT/pmc_ami_sq.txt" 2>&1
BENCH_ARGS="--no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_brk_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > "$OUT/pmc_brk_sq.txt" 2>&1
# profiles: kernel trace + PMC (separate passes)
bash scripts/profile_gpu.sh ${TAG} --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_space_invaders --game space_invaders --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_amidar --game amidar --no-extras > /dev/null 2>&1
for g in breakout space_invaders amidar; do bash scripts/profile_gpu.sh ${TAG}_${g}_4096 --game $g --envs 4096 --no-extras > /dev/null 2>&1; done
BENCH_ARGS="--game space_invaders --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_si_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" > "$OUT/pmc_si_sq.txt" 2>&1
BENCH_ARGS="--game space_invaders --envs 4096 --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_si_sq_4096 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES" > "$OUT/pmc_si_sq_4096.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_mixed" -- python3 $REPO/bench.py --game mixed --envs 32766 --with-gather --no-cpu-baseline --steps 50 --warmup 5 --repeats 2 > /dev/null 2>&1
for g in breakout space_invaders amidar gridworld; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_agent_$g" -- python3 $REPO/bench.py --protocol agent --deepmind --game $g --steps 60 --warmup 5 > /dev/null 2>&1
done
cd "$REPO"; python scripts/ab_step.py space_invaders toybox_amd/csrc/libtoybox_amd.so scripts/ab/lib_prev.so > "$OUT/step_only_space_invaders.txt" 2>&1
find gpurun_out -size +8M -delete; du -sh gpurun_out/$TAG gpurun_out/prof_${TAG}* | tail -20
