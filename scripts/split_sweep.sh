#!/bin/bash
# waves per frame of the RGB rasterisers (TBX_OPT_RENDER_SPLIT) at small batch sizes: one box, one call
cd ${GRAFT_REPO_ROOT:-.}
for n in ${1:-4096 8192}; do
    python scripts/render_probe.py breakout 3 $n 300 5 8 9 10 11 12 15 20
    python scripts/render_probe.py amidar 3 $n 300 5 7 8 9 10 12 13 25
    python scripts/render_probe.py space_invaders 3 $n 300 3 5 7 9 12 18
done
