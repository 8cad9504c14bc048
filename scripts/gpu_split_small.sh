#!/bin/bash
# waves-per-frame sweeps of the RGB rasterisers at small batches (TBX_OPT_RENDER_SPLIT), one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
L=toybox_amd/csrc/libtoybox_amd.so
{
timeout 250 python scripts/split_ab.py amidar 4096 "0,5,6,8,10,12,13,25" $L
timeout 250 python scripts/split_ab.py amidar 8192 "0,6,8,10,12,13" $L
timeout 250 python scripts/split_ab.py breakout 4096 "0,4,5,7,20" $L
timeout 250 python scripts/split_ab.py space_invaders 4096 "0,9,15,18" $L
} > gpurun_out/split_small.txt 2>&1
