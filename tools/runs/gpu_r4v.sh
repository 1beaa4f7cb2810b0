R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -25 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
