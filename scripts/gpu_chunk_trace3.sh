# why do rollout chunks of 1 024 envs lose under a record ring?  kernel-trace timelines of the one-launch chunk form with and without the K = 4 ring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chunktrace3
mkdir -p $O
: > $O/summary.txt
for cfg in "1024 4" "1024 0" "2048 4"; do
  set -- $cfg
  tag=breakout_$1_ring$2
  LO_OVERLAP=2 LO_CHUNKS=4 LO_GATHER=$2 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 $R/scripts/loop_once.py breakout $1 chunks 160 > $O/$tag.log 2>&1
  echo "== $tag (loop_once.py breakout $1 chunks, TBX_OPT_ROLLOUT_CHUNKS = 4, ring K = $2; last 30 dispatches)" >> $O/summary.txt
  python3 $R/scripts/trace_timeline.py $O/$tag 30 >> $O/summary.txt 2>&1
  find $O/$tag -size +4M -delete
done
