cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_batched_interventions.py tests/test_oracle_golden.py tests/test_preproc.py -x -q -m gpu -k "space_invaders or full_size or rollout or fuzz or render_step or synthetic or mixed or interventions or golden or pipelined" 2>&1 | tail -8 > gpurun_out/t4.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/si_step_trace -- python3 $R/scripts/loop_once.py space_invaders 65536 pair 40 > /dev/null 2>&1
python3 $R/scripts/trace_gaps.py $R/gpurun_out/si_step_trace 78 > $R/gpurun_out/si_step_gaps.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/si_step_trace2 -- python3 $R/scripts/loop_once.py space_invaders 65536 step 200 > /dev/null 2>&1
python3 $R/scripts/trace_gaps.py $R/gpurun_out/si_step_trace2 150 >> $R/gpurun_out/si_step_gaps.txt 2>&1
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $R/gpurun_out/si_step_pmc_$CTR -- python3 $R/scripts/loop_once.py space_invaders 65536 pair 12 > /dev/null 2>&1
done
python3 - <<'PY' >> $R/gpurun_out/si_step_gaps.txt
import csv,glob,collections,os
R=os.environ.get("GRAFT_REPO_ROOT")
agg=collections.defaultdict(list)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(R+"/gpurun_out/si_step_pmc_%s/*/*counter_collection.csv"%c):
        rows=list(csv.DictReader(open(f)))
        for r in rows[-40:]:
            k="step" if "si_step" in r["Kernel_Name"] else "render" if "render" in r["Kernel_Name"] else None
            if k: agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, "%.4g KiB avg over %d launches" % (sum(v)/len(v), len(v)))
PY
find $R/gpurun_out -name "*.csv" -size +2M -delete
