O=gpurun_out/r05t; mkdir -p $O
L=toybox_amd/csrc/libtoybox_amd.so
timeout 300 python scripts/epw_check.py > $O/parity.txt 2>&1
AB_PREROLL=300 timeout 300 python scripts/ab_step.py amidar $L $L:form3 > $O/ab_step.txt 2>&1
AB_STEP=1 AB_PREROLL=300 timeout 300 python scripts/ab_render.py amidar 3 $L $L:form3 > $O/ab_pair.txt 2>&1
AB_PREROLL=60 timeout 300 python scripts/ab_agent.py amidar $L $L:form3 $L:ring $L:ring:form3 > $O/ab_agent.txt 2>&1
AB_ENVS=16384 AB_PREROLL=300 timeout 300 python scripts/ab_step.py amidar $L:form2 $L:form1 $L:form3 > $O/ab_step_16384.txt 2>&1
AB_ENVS=32768 AB_PREROLL=300 timeout 300 python scripts/ab_step.py amidar $L:form2 $L:form1 $L:form3 > $O/ab_step_32768.txt 2>&1
tail -2 $O/parity.txt; tail -2 $O/ab_step.txt; tail -2 $O/ab_pair.txt; tail -4 $O/ab_agent.txt; tail -3 $O/ab_step_16384.txt; tail -3 $O/ab_step_32768.txt
