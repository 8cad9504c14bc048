O=gpurun_out/r05o; mkdir -p $O
for g in amidar gridworld; do
  AB_PREROLL=60 timeout 300 python scripts/ab_agent.py $g scripts/ab/lib_before_slotcopy.so toybox_amd/csrc/libtoybox_amd.so scripts/ab/lib_before_slotcopy.so:ring toybox_amd/csrc/libtoybox_amd.so:ring > $O/ab_agent_$g.txt 2>&1
done
timeout 1500 python -m pytest tests/test_preproc.py tests/test_gridworld.py tests/test_gpu_parity.py -x -q -m gpu -k "amidar or gridworld" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
