#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q -k "space_invaders or pipelined or frame_parity or level_transitions or fuzzed or mixed" > gpurun_out/s3/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s3/pytest.txt
tail -15 gpurun_out/s3/pytest.txt
for pre in 30 400; do
AB_PREROLL=$pre timeout 300 python scripts/ab_render.py space_invaders 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 2>&1 | tail -6
done
AB_ENVS=4096 AB_PREROLL=400 timeout 300 python scripts/ab_render.py space_invaders 3 scripts/ab/lib_prev.so toybox_amd/csrc/libtoybox_amd.so 2>&1 | tail -6
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from toybox_amd import Engine, _abi, hip
for n in (65536, 4096):
    e = Engine("space_invaders", n); e.seed(1234)
    for t in range(400): e.step_synthetic(1337, t)
    for split in (3, 5, 7, 9, 12, 18, 35):
        e.set_option(_abi.OPT_RENDER_SPLIT, split)
        e.render_device(channels=3); hip.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): e.render_device(channels=3)
        hip.synchronize(); dt = (time.perf_counter() - t0) / 30
        print("n %d split %2d  %.4f ms %.0f GB/s" % (n, split, dt * 1e3, n * 210 * 320 * 3 / dt / 1e9), flush=True)
    e.close()
PY
timeout 300 python bench.py --game space_invaders --no-cpu-baseline > gpurun_out/s3/bench_si.json 2> gpurun_out/s3/bench_si.err; python -c "
import json; d=json.load(open('gpurun_out/s3/bench_si.json')); print(d['value'], d['ms_per_step'], d['pipeline'], d['roofline']['frac'], d.get('serialised',{}).get('ms_per_step'), d.get('serialised',{}).get('roofline_frac'), d.get('scaling_strong',{}).get('share_of_linear'))"
