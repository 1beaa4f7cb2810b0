R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention_exact" > $O/pytest_attn.log 2>&1; echo "pytest attn rc=$?"
tail -3 $O/pytest_attn.log
timeout 300 python tools/attn_bench.py 20 3 10,14 extra > $O/attn_variants.txt 2>&1; echo "attn rc=$?"
cat $O/attn_variants.txt
