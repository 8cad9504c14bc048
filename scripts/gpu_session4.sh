#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
(cd scripts/ubench && ./write_bw6 | tail -15)
AB_PREROLL=400 timeout 300 python scripts/ab_render.py space_invaders 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 2>&1 | tail -6
AB_PREROLL=400 timeout 300 python scripts/ab_render.py breakout 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 2>&1 | tail -4
AB_PREROLL=400 timeout 300 python scripts/ab_render.py amidar 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 2>&1 | tail -4
