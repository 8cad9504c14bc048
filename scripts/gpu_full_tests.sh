cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
