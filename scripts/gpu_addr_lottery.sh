# does the rate of the overlapped rollout forms depend on WHERE the engine's buffers lie?  the same loop (scripts/box_probe.py) in processes
# that differ only in the size of one allocation made before the engine exists
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/probe
mkdir -p $O
F=$O/addr_$(date +%s).txt
python3 $R/scripts/box_probe.py 8192 100 0 | grep "^box id" > $F
for n in ${AL_SIZES:-8192 16384}; do
  for kb in ${AL_SHIFTS:-0 4 64 1024 2052 65536 1048576 3145732}; do
    echo "== $n envs, BP_SHIFT_KB=$kb" >> $F
    BP_CLOCK=0 BP_ADDR=1 BP_SHIFT_KB=$kb timeout 120 python3 $R/scripts/box_probe.py $n $((8192 * 8000 / n)) ${AL_ROUNDS:-2} 2>&1 | grep "^round\|^addresses" >> $F
  done
done
cat $F
