for n in 1024 2048 8192; do echo "== $n envs, ring K=4"; BP_CLOCK=0 BP_FORMS=order,4 timeout 120 python3 scripts/box_probe.py $n 16000 1 2>&1 | grep "^round"; done
echo "== 1024 envs, no gather"; BP_CLOCK=0 BP_GATHER=0 BP_FORMS=order,4 timeout 120 python3 scripts/box_probe.py 1024 16000 1 2>&1 | grep "^round"
