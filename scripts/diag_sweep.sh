#!/bin/bash
# stores-only / painting-only SpaceInvaders render launches (TBX_SI_DIAG: 1 = no painting, 2 = no stores, 3 = neither) against
# the full kernel, per waves-per-frame value.  Needs a MEASUREMENT build of the library: make -C toybox_amd/csrc clean all DIAG=1
# (the shipped build ignores TBX_SI_DIAG).  usage: diag_sweep.sh "<diag values>" "<splits>" [envs]
cd ${GRAFT_REPO_ROOT:-.}
for d in ${1:-0 1 2 3}; do TBX_SI_DIAG=$d python scripts/render_probe.py space_invaders 3 ${3:-65536} 400 ${2:-1 5 9}; done
