O=gpurun_out/r05q; mkdir -p $O
AB_PREROLL=60 timeout 300 python scripts/ab_agent.py amidar scripts/ab/lib_before_dense_runs.so toybox_amd/csrc/libtoybox_amd.so scripts/ab/lib_before_dense_runs.so:ring toybox_amd/csrc/libtoybox_amd.so:ring > $O/ab_agent_amidar.txt 2>&1
AB_PREROLL=60 timeout 400 python scripts/ab_agent.py gridworld scripts/ab/lib_before_dense_runs.so toybox_amd/csrc/libtoybox_amd.so scripts/ab/lib_gw4.so scripts/ab/lib_before_dense_runs.so:ring toybox_amd/csrc/libtoybox_amd.so:ring scripts/ab/lib_gw4.so:ring > $O/ab_agent_gridworld.txt 2>&1
timeout 1500 python -m pytest tests/test_preproc.py tests/test_gridworld.py tests/test_gpu_parity.py -x -q -m gpu -k "amidar or gridworld" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
