# kernel-trace timelines of rollout chunks at the sizes where they gain (8 192, 65 536 envs) and where they lose (16 384) on one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chunktrace2
mkdir -p $O
python3 $R/scripts/box_probe.py 8192 100 0 | grep "^box id" > $O/summary.txt
for cfg in "breakout 8192 chunks 0" "breakout 16384 chunks 0" "breakout 65536 chunks 0"; do
  set -- $cfg
  tag=$1_$2_$3
  LO_OVERLAP=$4 LO_CHUNKS=1 LO_GATHER=4 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 $R/scripts/loop_once.py $1 $2 $3 160 > $O/$tag.log 2>&1
  echo "== $tag (loop_once.py $1 $2 $3, K = 4 ring; last 24 dispatches)" >> $O/summary.txt
  python3 $R/scripts/trace_timeline.py $O/$tag 34 >> $O/summary.txt 2>&1
  find $O/$tag -size +4M -delete
done
