# kernel trace of bench.py's own fused loop at 8 192 envs with the K = 4 ring, overlapped launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ovtrace2
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/bench -- python3 $R/bench.py --envs 8192 --with-gather --fused-overlap on --steps 60 --repeats 2 --no-configs --no-extras --no-cpu-baseline --preroll 100 --settle 10 --warmup 5 > $O/bench.log 2>&1
python3 $R/scripts/trace_timeline.py $O/bench 60 > $O/summary.txt 2>&1
LO_OVERLAP=1 LO_GATHER=4 rocprofv3 --kernel-trace --output-format csv -d $O/loop -- python3 $R/scripts/loop_once.py breakout 8192 fused 60 > $O/loop.log 2>&1
python3 $R/scripts/trace_timeline.py $O/loop 40 >> $O/summary.txt 2>&1
find $O -size +4M -delete
