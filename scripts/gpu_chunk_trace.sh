# kernel-trace timelines of the rollout loop at 4 096 and 8 192 Breakout envs (K = 4 ring): stream order against rollout chunks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chunktrace
mkdir -p $O
for cfg in "breakout 4096 fused 2" "breakout 4096 chunks 0" "breakout 8192 chunks 0" "space_invaders 4096 chunks 0"; do
  set -- $cfg
  tag=$1_$2_$3
  LO_OVERLAP=$4 LO_GATHER=4 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 $R/scripts/loop_once.py $1 $2 $3 80 > $O/$tag.log 2>&1
  echo "== $tag (loop_once.py $1 $2 $3, K = 4 ring; last 30 dispatches)" >> $O/summary.txt
  python3 $R/scripts/trace_timeline.py $O/$tag 30 >> $O/summary.txt 2>&1
  find $O/$tag -size +4M -delete
done
