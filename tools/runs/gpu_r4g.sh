R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
