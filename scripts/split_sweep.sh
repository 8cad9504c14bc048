#!/bin/bash
# render launch time against waves-per-frame (`split`) at small batches, one process per setting (scripts/render_probe.py)
# usage (GPU box): bash scripts/split_sweep.sh "4096 8192" > gpurun_out/split_sweep.txt
for n in ${1:-4096 8192}; do
  for sp in 0 1 2 3 4 5 6 8 10 12 15; do
    TBX_BRK_SPLIT=$sp python scripts/render_probe.py breakout 3 $n 300
    TBX_RENDER_SPLIT=$sp python scripts/render_probe.py amidar 3 $n 300
    TBX_RENDER_SPLIT=$sp python scripts/render_probe.py space_invaders 3 $n 300
  done
done
