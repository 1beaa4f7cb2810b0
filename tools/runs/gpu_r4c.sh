R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q > $O/pytest_fp8.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fp8.log; tail -40 $O/pytest_fp8.log
