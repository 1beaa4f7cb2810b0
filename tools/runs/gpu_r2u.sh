R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python tools/attn_bench.py 20 4 0,4 > $O/ab.log 2>&1; grep "vit_d64" $O/ab.log
