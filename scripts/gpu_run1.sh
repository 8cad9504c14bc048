set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_paths.py -x -q -m gpu -k "render_step or ring or rewritten_state or gather or pipelined" 2>&1 | tail -15 > gpurun_out/t1.log
timeout 600 python scripts/strong_sweep.py breakout 8192 65536 4096 > gpurun_out/sweep1.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench1.log 2>&1
