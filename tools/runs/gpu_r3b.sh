# round 3: first run of attn2_kernel (D = 64 non-causal): parity tests, then A/B against the round-2 kernel (variant 5)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
tail -15 $O/pytest_attn.log
timeout 300 python tools/attn_bench.py 20 3 0,5 > $O/attn_bench.txt 2>&1
cat $O/attn_bench.txt
