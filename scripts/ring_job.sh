O=gpurun_out/r05n; mkdir -p $O
for g in breakout space_invaders amidar gridworld; do
  AB_PREROLL=60 timeout 300 python scripts/ab_agent.py $g scripts/ab/lib_before_ring.so toybox_amd/csrc/libtoybox_amd.so toybox_amd/csrc/libtoybox_amd.so:ring > $O/ab_agent_ring_$g.txt 2>&1
done
timeout 1200 python -m pytest tests/test_preproc.py tests/test_gpu_paths.py -x -q -m gpu -k "ring or plane or buffer or fused_observation" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
