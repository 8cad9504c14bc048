#!/bin/bash
# stores-only / painting-only SpaceInvaders render launches (TBX_SI_DIAG: 1 = no painting, 2 = no stores, 3 = neither) against
# waves per frame: what bounds the rasteriser (the Amidar figures there came from a temporary build
# with the same switch).
# usage (GPU box): bash scripts/diag_sweep.sh "0 1 2 3" "1 5 9" [envs]
for sp in ${2:-1 5 9}; do for d in ${1:-0 1 2 3}; do TBX_SI_DIAG=$d TBX_RENDER_SPLIT=$sp python scripts/render_probe.py space_invaders 3 ${3:-65536} 400; done; done
