R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ah; mkdir -p $O
cd $R
for rep in 1 2 3; do
MJV_BENCH_ROUNDS=1 timeout 600 python tools/gemm_bench.py 1000 2>&1 | grep -E "vit_fc1 "
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gelu" 2>&1 | tail -2
