#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]; output under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== host: $(nproc) nproc, affinity $(python3 -c 'import os;print(len(os.sched_getaffinity(0)))'), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" | tee "$OUT/host.txt"
BENCH="python3 $REPO/bench.py --no-cpu-baseline $*"
# PROFILE_STEPS / PROFILE_WARMUP: the step counts of the traced run (default 100 / 10; "20 5" = the driver's own command)
TS=${PROFILE_STEPS:-100}; TW=${PROFILE_WARMUP:-10}
echo "bench.py --no-cpu-baseline $* --steps $TS --warmup $TW" > "$OUT/cmd.txt"
# the sources these counters are read on (bench.py reports a traffic figure measured on other sources as stale)
python3 -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.csrc_fingerprint())" > "$OUT/csrc_sha16.txt" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps $TS --warmup $TW > "$OUT/trace.log" 2>&1
tail -2 "$OUT/trace.log"
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT/pmc_$CTR" -- $BENCH --steps 10 --warmup 2 > "$OUT/pmc_$CTR.log" 2>&1
  tail -1 "$OUT/pmc_$CTR.log" | cut -c1-200
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_misc" -- $BENCH --steps 10 --warmup 2 > "$OUT/pmc_misc.log" 2>&1
find "$OUT" -name "*.csv" | head -40
# keep the merge-back small: drop anything big
find "$OUT" -size +8M -delete
du -sh "$OUT"
