set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_paths.py tests/test_batched_interventions.py -x -q -m gpu -k "render_step or interventions or helpers" 2>&1 | tail -15 > gpurun_out/t2.log
timeout 600 python scripts/strong_sweep.py space_invaders 4096 65536 > gpurun_out/sweep2.log 2>&1
timeout 300 python bench.py --game space_invaders --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench2_si.log 2>&1
timeout 300 python bench.py --game space_invaders --envs 4096 --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/bench2_si4096.log 2>&1
timeout 300 python bench.py --envs 4096 --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/bench2_brk4096.log 2>&1
