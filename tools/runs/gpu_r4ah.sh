R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ah; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_preprocess_gpu.py -m gpu -q -x > $O/pytest_pre.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pre.log; tail -4 $O/pytest_pre.log
timeout 300 python tools/preprocess_bench.py 2>/dev/null | tee $O/preprocess_bench.txt
