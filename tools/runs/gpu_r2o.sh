R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "split_k" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
MJV_BENCH_TAILS=1 timeout 600 python tools/gemm_bench.py 9000 9004 9008 9016 > $O/tails.log 2>&1; grep -v "^tail_vit\|^main" $O/tails.log
