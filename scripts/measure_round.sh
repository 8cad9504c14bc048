#!/bin/bash
# Everything DESIGN.md section 6 quotes, measured in ONE gpurun call on ONE box (boxes differ by up to ~15 %, so numbers of
# different calls are not comparable): bench lines per game / batch size / protocol, then rocprofv3 kernel-trace + PMC passes.
# usage (on the GPU box): bash scripts/measure_round.sh r02
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline"
python bench.py > "$OUT/bench_breakout_65536.json" 2> "$OUT/bench_breakout_65536.err"
for g in space_invaders amidar gridworld; do $B --game $g --no-extras > "$OUT/bench_${g}_65536.json" 2>&1; done
for g in breakout space_invaders amidar; do $B --game $g --envs 4096 --no-extras > "$OUT/bench_${g}_4096.json" 2>&1; done
$B --game mixed --envs 32766 --with-gather > "$OUT/bench_mixed_32766.json" 2>&1
$B --envs 8192 --with-gather --no-extras > "$OUT/bench_breakout_8192_gather.json" 2>&1
$B --with-gather --no-extras > "$OUT/bench_breakout_65536_gather.json" 2>&1
for g in breakout space_invaders amidar gridworld; do
  python bench.py --protocol agent --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}.json" 2>&1
  python bench.py --protocol agent --deepmind --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}_deepmind.json" 2>&1
  python bench.py --protocol reference --game $g > "$OUT/reference_${g}.json" 2>&1
done
python scripts/host_latency.py > "$OUT/host_latency.txt" 2>&1
# profiles: kernel trace + PMC (separate passes)
bash scripts/profile_gpu.sh ${TAG} --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_space_invaders --game space_invaders --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_amidar --game amidar --no-extras > /dev/null 2>&1
for g in breakout space_invaders amidar; do bash scripts/profile_gpu.sh ${TAG}_${g}_4096 --game $g --envs 4096 --no-extras > /dev/null 2>&1; done
BENCH_ARGS="--game space_invaders --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_si_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" > "$OUT/pmc_si_sq.txt" 2>&1
BENCH_ARGS="--game space_invaders --envs 4096 --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_si_sq_4096 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES" > "$OUT/pmc_si_sq_4096.txt" 2>&1
BENCH_ARGS="--game breakout --envs 4096 --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_brk_sq_4096 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES" > "$OUT/pmc_brk_sq_4096.txt" 2>&1
BENCH_ARGS="--game amidar --envs 4096 --no-extras --repeats 1" bash scripts/pmc_gpu.sh ${TAG}_ami_sq_4096 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES" > "$OUT/pmc_ami_sq_4096.txt" 2>&1
for g in breakout space_invaders amidar gridworld; do
  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_agent_$g" -- python3 $REPO/bench.py --protocol agent --deepmind --game $g --steps 60 --warmup 5 > /dev/null 2>&1
done
cd "$REPO"; python scripts/ab_step.py space_invaders toybox_amd/csrc/libtoybox_amd.so > "$OUT/step_only_space_invaders.txt" 2>&1; python scripts/ab_step.py amidar toybox_amd/csrc/libtoybox_amd.so > "$OUT/step_only_amidar.txt" 2>&1; AB_ENVS=4096 python scripts/ab_step.py amidar toybox_amd/csrc/libtoybox_amd.so > "$OUT/step_only_amidar_4096.txt" 2>&1
find gpurun_out -size +8M -delete; du -sh gpurun_out/$TAG gpurun_out/prof_${TAG}* | tail -20
