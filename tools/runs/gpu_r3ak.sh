R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ak; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "rising" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
