#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s1
timeout 400 python scripts/pipeline_sweep.py breakout 4096 8192 12288 16384 32768 65536 > gpurun_out/s1/sweep.txt 2>&1
PS_GATHER=1 timeout 300 python scripts/pipeline_sweep.py breakout 4096 8192 16384 > gpurun_out/s1/sweep_gather.txt 2>&1
GPU_MAX_HW_QUEUES=8 PS_GATHER=1 timeout 200 python scripts/pipeline_sweep.py breakout 4096 8192 > gpurun_out/s1/sweep_gather_q8.txt 2>&1
cat gpurun_out/s1/sweep.txt; echo gather; cat gpurun_out/s1/sweep_gather.txt; echo q8; cat gpurun_out/s1/sweep_gather_q8.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s1/pytest.txt
tail -5 gpurun_out/s1/pytest.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err; tail -c 1500 gpurun_out/s1/bench.json; tail -3 gpurun_out/s1/bench.err
