R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ac; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_preprocess_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
