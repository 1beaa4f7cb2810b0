R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ao; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "fp8 or mxfp8 or sweep or gemm" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 400 python tools/fuzz_kernels.py 240 99 2>&1 | tail -2
