cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gaps
for cfg in "amidar 65536 pair 40" "amidar 65536 render 40" "amidar 4096 pair 200" "breakout 65536 pair 40" "breakout 65536 fused 40" "breakout 8192 fused 200" "space_invaders 65536 pair 40"; do
  set -- $cfg
  tag=$1_$2_$3
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps/$tag -- python3 $R/scripts/loop_once.py $1 $2 $3 $4 > $R/gpurun_out/gaps/$tag.log 2>&1
  echo "== $tag" >> $R/gpurun_out/gaps/summary.txt
  python3 $R/scripts/trace_gaps.py $R/gpurun_out/gaps/$tag $(( $4 * 2 - 2 )) >> $R/gpurun_out/gaps/summary.txt 2>&1
  find $R/gpurun_out/gaps/$tag -size +4M -delete
done
