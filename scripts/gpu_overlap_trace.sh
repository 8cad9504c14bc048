# kernel traces of the fused loop, stream order against overlapped (TBX_OPT_FUSED_OVERLAP), with and without the K = 4 record ring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ovtrace
mkdir -p $O
for cfg in "8192 2 0" "8192 1 0" "8192 1 4" "8192 2 4" "4096 1 4" "65536 1 0"; do
  set -- $cfg
  tag=n$1_ov$2_k$3
  LO_OVERLAP=$2 LO_GATHER=$3 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 $R/scripts/loop_once.py breakout $1 fused 60 > $O/$tag.log 2>&1
  echo "== $tag" >> $O/summary.txt
  python3 $R/scripts/trace_timeline.py $O/$tag 26 >> $O/summary.txt 2>&1
  find $O/$tag -size +4M -delete
done
