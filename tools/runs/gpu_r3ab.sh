R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "random_batch" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "random batches|passed|failed|Error|assert" $O/pytest.log | tail -12
