R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O
cd $R
timeout 900 python tools/attn_bench.py 20 3 0,7 x > $O/attn_bench.txt 2>&1; grep -E "L8192|L28810|L2048" $O/attn_bench.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "long_context" 2>&1 | tail -3
