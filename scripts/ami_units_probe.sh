O=gpurun_out/r05g; mkdir -p $O
python - > $O/parity.txt 2>&1 <<'PY'
import ctypes, sys, numpy as np
sys.path.insert(0, '.')
from toybox_amd import Engine, _abi
olib = ctypes.CDLL('oracle/liboracle.so'); _abi.bind(olib)
for name in ('scripts/ab/lib_ami_rows12.so', 'scripts/ab/lib_ami_rows8.so'):
    lib = _abi.bind(ctypes.CDLL(name))
    for split in (0, 1, 5, 7, 11, 16):
        g, o = Engine('amidar', 700, lib=lib), Engine('amidar', 700, lib=olib)
        for e in (g, o):
            e.seed(5); e.new_game()
        if split: g.set_option(_abi.OPT_RENDER_SPLIT, split)
        for t in range(150):
            g.step_synthetic(1337, t); o.step_synthetic(1337, t)
        ok = all(np.array_equal(g.render(c), o.render(c)) for c in (3, 1, 4))
        print(name, 'split', split, 'frames equal:', ok)
        g.close(); o.close()
PY
cat $O/parity.txt
L="toybox_amd/csrc/libtoybox_amd.so scripts/ab/lib_ami_rows12.so scripts/ab/lib_ami_rows8.so"
for SP in "9,7,8" "9,5,11" "9,11,16" "9,6,10"; do echo "splits $SP" >> $O/ab.txt; AB_SPLIT=$SP AB_PREROLL=400 timeout 300 python scripts/ab_render.py amidar 3 $L >> $O/ab.txt 2>&1; done
cat $O/ab.txt
