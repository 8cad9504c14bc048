#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s6
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_envs.py tests/test_toybox_surface.py -m gpu -x -q -k "step1 or envs or surface or base_env or seed_repro" > gpurun_out/s6/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s6/pytest.txt
tail -12 gpurun_out/s6/pytest.txt
for g in breakout space_invaders amidar; do
  timeout 300 python bench.py --protocol reference --gym --game $g --reps 5 --steps 3000 > gpurun_out/s6/ref_$g.json 2> gpurun_out/s6/ref_$g.err
  python -c "
import json; d=json.load(open('gpurun_out/s6/ref_$g.json')); print('$g raw %.0f +- %.0f  gym %.0f +- %.0f  cpu raw %.0f gym %.0f' % (d['value'], d['sem'], d['gym']['value'], d['gym']['sem'], d['cpu_baseline']['value'], d['cpu_baseline']['gym']['value']))"
done
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from toybox_amd import Engine
for game in ("breakout", "space_invaders", "amidar"):
    e = Engine(game, 1); e.seed(3); e.new_game()
    a = e.legal_actions
    for ch in (1, 3):
        for t in range(200): e.step1_frame(0, a[t % len(a)], ch, auto_reset=True)
        t0 = time.perf_counter()
        for t in range(3000): e.step1_frame(0, a[t % len(a)], ch, auto_reset=True)
        dt = (time.perf_counter() - t0) / 3000
        print("%s step1_frame ch=%d: %.1f us per call" % (game, ch, dt * 1e6), flush=True)
    e.close()
PY
AB_PREROLL=400 timeout 300 python scripts/ab_render.py breakout 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 2>&1 | tail -6
