# chunk length without a gather at BASELINE's 4 096-env configs: order = the loop of single calls in stream order, 3 = a rasteriser launch per frame on two lanes, 4 = one per chunk
for g in breakout space_invaders; do
for K in 4 8 16; do echo "== $g 4096 envs, no gather, chunks of $K"; BP_GAME=$g BP_CLOCK=0 BP_GATHER=0 BP_K=$K BP_FORMS=order,3,4 timeout 120 python3 scripts/box_probe.py 4096 16000 2 2>&1 | grep "^round" | cut -c1-40; done
done
for K in 8 16; do echo "== breakout 2048 envs, no gather, chunks of $K"; BP_CLOCK=0 BP_GATHER=0 BP_K=$K BP_FORMS=order,3,4 timeout 120 python3 scripts/box_probe.py 2048 16000 1 2>&1 | grep "^round" | cut -c1-40; done
python3 scripts/box_probe.py 8192 100 0 | grep "^box id"
