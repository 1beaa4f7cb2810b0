R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3al; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "every_bf16" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
