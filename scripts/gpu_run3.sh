set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_paths.py -x -q -m gpu -k "render_step or pipelined" 2>&1 | tail -15 > gpurun_out/t3.log
timeout 600 python scripts/strong_sweep.py amidar 4096 16384 65536 > gpurun_out/sweep3.log 2>&1
timeout 300 python bench.py --game amidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench3_ami.log 2>&1
timeout 300 python bench.py --game amidar --envs 4096 --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/bench3_ami4096.log 2>&1
