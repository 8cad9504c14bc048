#!/bin/bash
# the PCIe-inclusive rates (bench.py --protocol host): frames and agent observations delivered to host memory
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for g in breakout space_invaders amidar; do timeout 300 python bench.py --protocol host --game $g --envs 8192 --steps 20 --warmup 3; done
timeout 300 python bench.py --protocol host --game breakout --envs 32768 --steps 8 --warmup 2
} 2>/dev/null | grep '^{' > gpurun_out/host_rate.txt
timeout 300 python -m pytest tests/test_envs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/t12.log
