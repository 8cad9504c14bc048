#!/bin/bash
# stores-only / painting-only render launches (TBX_*_DIAG) against waves per frame: what bounds a rasteriser
# usage (GPU box): bash scripts/diag_sweep.sh amidar "0 1 2 3" "1 5 9 12"
game=$1; var=TBX_AMI_DIAG; [ "$game" = space_invaders ] && var=TBX_SI_DIAG
for sp in ${3:-0}; do for d in ${2:-0 1 2 3}; do env $var=$d TBX_RENDER_SPLIT=$sp python scripts/render_probe.py $game 3 ${4:-65536} 400; done; done
