O=gpurun_out/r05p; mkdir -p $O
AB_PREROLL=60 timeout 300 python scripts/ab_agent.py breakout scripts/ab/lib_before_runs.so toybox_amd/csrc/libtoybox_amd.so scripts/ab/lib_before_runs.so:ring toybox_amd/csrc/libtoybox_amd.so:ring > $O/ab_agent_breakout.txt 2>&1
AB_DEEPMIND=1 AB_PREROLL=60 timeout 300 python scripts/ab_agent.py breakout scripts/ab/lib_before_runs.so toybox_amd/csrc/libtoybox_amd.so > $O/ab_agent_breakout_dm.txt 2>&1
timeout 1500 python -m pytest tests/test_preproc.py tests/test_gpu_parity.py tests/test_envs.py -x -q -m gpu -k "breakout" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
