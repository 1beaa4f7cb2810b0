R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "single_layer or ntk or mjbench or two_threads or k_sliced or trimming" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "vs reference|third call|distance to|K-sliced|passed|failed|rc=|Error" $O/pytest.log | tail -30
