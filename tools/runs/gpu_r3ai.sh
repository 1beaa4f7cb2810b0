R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ai; mkdir -p $O
cd $R
MJV_BENCH_ROUNDS=2 timeout 900 python tools/gemm_bench.py 256 128 2>&1 | grep -E "vit_|llm_wo|square"
