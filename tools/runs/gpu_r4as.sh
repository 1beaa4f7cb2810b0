R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4as; mkdir -p $O
cd $R
MJV_BENCH_TAILS=1 MJV_BENCH_WSTD=0.02 MJV_BENCH_ROUNDS=5 timeout 600 python tools/gemm_bench.py 5000 4432 4416 2>/dev/null | grep "b1_llm\|tail_llm\|b1_vit" | tee $O/split_gate.txt
