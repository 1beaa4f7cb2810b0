R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
for rep in 1 2 3; do
for lib in bench gs; do
MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_$lib.so MJV_BENCH_ROUNDS=1 timeout 600 python tools/gemm_bench.py 1000 2>&1 | grep -E "vit_fc1 |vit_qkv" | sed "s/^/$lib /"
done; done
MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_gs.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gelu" 2>&1 | tail -2
