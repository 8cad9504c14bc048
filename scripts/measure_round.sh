#!/bin/bash
# Everything DESIGN.md section 6 quotes, measured in ONE gpurun call on ONE box (boxes differ by up to ~15 %, so numbers of
# different calls are not comparable): bench lines per game / batch size / protocol, the N-process dress rehearsal on one GPU,
# loop-form sweeps, rocprofv3 kernel-trace + PMC passes, rasterisers against the previous round's build.
# usage (on the GPU box): bash scripts/measure_round.sh r06        then, back home: python scripts/collect_round.py r06
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline"
# ---- bench lines.  The first one is the driver's own command: it carries serialised, step_only, scaling_strong, BASELINE
# configs 2-5, cpu_baseline, cpu_config1
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_breakout_65536.json" 2> "$OUT/bench_breakout_65536.err"
$B --no-configs > "$OUT/bench_breakout_65536_steps200.json" 2>/dev/null
for g in space_invaders amidar; do $B --game $g > "$OUT/bench_${g}_65536.json" 2>/dev/null; done
$B --game gridworld --no-extras > "$OUT/bench_gridworld_65536.json" 2>/dev/null
for g in breakout space_invaders amidar; do $B --game $g --envs 4096 --steps 400 > "$OUT/bench_${g}_4096.json" 2>/dev/null; done
python bench.py --game mixed --envs 32768 --with-gather --cpu-seconds 8 > "$OUT/bench_mixed_32768.json" 2>/dev/null
$B --game mixed --envs 32768 --with-gather --loop pair --gather-every 1 > "$OUT/bench_mixed_32768_pair_k1.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 > "$OUT/bench_breakout_8192_gather.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 --loop pair > "$OUT/bench_breakout_8192_gather_pair_k4.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 --loop pair --gather-every 1 > "$OUT/bench_breakout_8192_gather_pair_k1.json" 2>/dev/null
# ---- the N-process flow on ONE GPU: 2 and 8 ranks on device 0, host transport of the gather (no RCCL between two ranks of one device)
python bench.py --gpus 2 --gather host --one-device --cpu-seconds 4 > "$OUT/rehearsal_2ranks_host_transport.json" 2> "$OUT/rehearsal_2ranks.err"
python bench.py --gpus 8 --gather host --one-device --cpu-seconds 4 --steps 20 --warmup 5 > "$OUT/rehearsal_8ranks_host_transport.json" 2> "$OUT/rehearsal_8ranks.err"
# ... and under the driver's own launch line for N > 1 (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --gather host --one-device --cpu-seconds 4 > "$OUT/rehearsal_2ranks_torch_distributed_run.json" 2> "$OUT/rehearsal_2ranks_tdr.err"
for g in breakout space_invaders amidar gridworld; do
  python bench.py --protocol agent --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}.json" 2>/dev/null
  python bench.py --protocol agent --deepmind --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}_deepmind.json" 2>/dev/null
  python bench.py --protocol agent --obs ring --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}_ring.json" 2>/dev/null
  python bench.py --protocol agent --obs ring --deepmind --game $g --steps 100 --warmup 10 > "$OUT/agent_${g}_ring_deepmind.json" 2>/dev/null
done
for g in breakout space_invaders amidar; do
  python bench.py --protocol reference --gym --game $g --reps 10 > "$OUT/reference_${g}.json" 2>/dev/null        # 10 reps x 10 000 steps, both arms, CPU beside
done
for g in breakout space_invaders amidar; do
  python bench.py --protocol host --game $g --envs 8192 --steps 20 --warmup 3 > "$OUT/host_${g}.json" 2>/dev/null        # PCIe-inclusive: frames / observations to host memory
done
python bench.py --protocol host --game breakout --envs 64 --steps 200 --warmup 20 > "$OUT/host_breakout_64envs.json" 2>/dev/null
# ---- loop forms and ring depths against each other, interleaved in one process
timeout 400 python scripts/strong_sweep.py breakout 4096 8192 65536 2>&1 | grep '^{' > "$OUT/sweep_breakout.txt"
# ---- round 6: the random-rollout loop in stream order / overlapped behind the ticket / as rollout chunks, ONE engine per process
for g in 1 0; do RA_GATHER=$g timeout 300 python scripts/rollout_ab.py 2048 4096 8192 16384 65536 2>&1 | grep '^{' >> "$OUT/rollout_forms.txt"; done
$B --envs 4096 --steps 400 --rollout-chunks off --fused-overlap off --no-extras > "$OUT/bench_breakout_4096_stream_order.json" 2>/dev/null
$B --envs 4096 --steps 400 --rollout-chunks off --fused-overlap on --no-extras > "$OUT/bench_breakout_4096_ticket.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 --rollout-chunks on > "$OUT/bench_breakout_8192_gather_chunks.json" 2>/dev/null
$B --envs 8192 --with-gather --no-extras --steps 400 --rollout-chunks off > "$OUT/bench_breakout_8192_gather_stream_order.json" 2>/dev/null
# ... the placement lottery of side-by-side rasteriser launches on THIS box (order / per frame on two lanes / one launch per chunk, processes that
# differ in one allocation in front of the engine), chunk lengths, and the shader clock while the loops run
BP_FORMS=order,3,4 AL_ROUNDS=1 AL_SIZES="8192 16384" AL_SHIFTS="0 4 2052 1048576" bash scripts/gpu_addr_lottery.sh > "$OUT/addr_lottery.txt" 2>&1
bash scripts/gpu_chunk_lengths.sh > "$OUT/chunk_lengths.txt" 2>&1
make -C scripts/ubench libclock_probe.so > /dev/null 2>&1; BP_FORMS=order,3,4 timeout 200 python scripts/box_probe.py 8192 16000 2 2>&1 | grep "^box id\|^round\|VBIOS\|^0 " > "$OUT/box_probe.txt"
# ... and this build against the previous round's on the headline loop (stream order at 65 536 envs)
timeout 300 python scripts/fused_lib_ab.py scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 65536 4096 2>&1 | grep '^{' > "$OUT/fused_lib_ab.txt"
# ---- the rasterisers (and [step ; render]) against the previous round's build, interleaved
for g in breakout space_invaders amidar; do
  AB_PREROLL=400 timeout 300 python scripts/ab_render.py $g 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so > "$OUT/ab_render_$g.txt" 2>&1
done
for g in breakout space_invaders amidar gridworld; do
  AB_PREROLL=60 timeout 300 python scripts/ab_agent.py $g scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so > "$OUT/ab_agent_$g.txt" 2>&1
done
# ---- the agent observation kernel of SpaceInvaders taken apart (DIAG build) and the store-alignment microbenchmark
for g in space_invaders amidar; do
  bash scripts/agent_diag.sh $g 65536 > /dev/null 2>&1
  cp "$REPO/gpurun_out/agent_diag/times_$g.txt" "$OUT/agent_diag_times_$g.txt" 2>/dev/null
  cp "$REPO/gpurun_out/agent_diag/counters_$g.txt" "$OUT/agent_diag_counters_$g.txt" 2>/dev/null
  AGENT_DIAG_OBS=ring timeout 400 python scripts/agent_diag.py $g 65536 > "$OUT/agent_diag_times_${g}_ring.txt" 2>&1
done
make -C scripts/ubench write_align > /dev/null 2>&1; timeout 200 scripts/ubench/write_align > "$OUT/write_align.txt" 2>&1
# ---- profiles: kernel trace + PMC (separate passes).  The first one is the driver's own command (--steps 20 --warmup 5).
PROFILE_STEPS=20 PROFILE_WARMUP=5 bash scripts/profile_gpu.sh ${TAG} --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_pair --no-extras --loop pair --pipeline 0 > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_space_invaders --game space_invaders --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_amidar --game amidar --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_breakout_4096 --envs 4096 --no-extras > /dev/null 2>&1
for g in breakout space_invaders amidar; do bash scripts/profile_gpu.sh ${TAG}_${g}_4096_pair --game $g --envs 4096 --no-extras --loop pair --pipeline 0 > /dev/null 2>&1; done
bash scripts/profile_gpu.sh ${TAG}_breakout_8192_gather --envs 8192 --with-gather --no-extras > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for g in breakout space_invaders amidar gridworld; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_agent_$g" -- python3 $REPO/bench.py --protocol agent --deepmind --game $g --steps 60 --warmup 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_agent_ring_$g" -- python3 $REPO/bench.py --protocol agent --deepmind --obs ring --game $g --steps 60 --warmup 5 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_mixed/trace" -- python3 $REPO/bench.py --game mixed --envs 32768 --with-gather --no-cpu-baseline --steps 50 --warmup 5 --repeats 2 > /dev/null 2>&1
cd "$REPO"
find gpurun_out -size +8M -delete; du -sh gpurun_out/$TAG gpurun_out/prof_${TAG}* 2>/dev/null | tail -20
