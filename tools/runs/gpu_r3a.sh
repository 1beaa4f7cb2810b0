# round 3 baseline: GPU tests, default bench line, attention + GEMM micro-benchmarks on the unchanged round-2 build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
timeout 300 python tools/attn_bench.py 20 3 0 full > $O/attn_bench.txt 2>&1
timeout 600 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1
tail -3 $O/pytest.log; tail -c 1500 $O/bench_default.json; cat $O/attn_bench.txt | tail -20; tail -30 $O/gemm_bench.txt
