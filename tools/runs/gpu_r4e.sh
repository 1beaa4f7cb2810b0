R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -s -k "single_layer_mxfp8 or c1_dims or rank_agreement" > $O/pytest_fp8_big.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fp8_big.log; grep -v "^$" $O/pytest_fp8_big.log | tail -30
