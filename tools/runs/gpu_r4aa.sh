R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4aa; mkdir -p $O
cd $R
timeout 700 python tools/fuzz_kernels.py 420 20261003 > $O/fuzz_420s.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_420s.txt; tail -4 $O/fuzz_420s.txt
timeout 300 python tools/gemm_stress.py 100 > $O/gemm_stress.txt 2>&1; echo "gemm_stress rc=$?" >> $O/gemm_stress.txt; tail -3 $O/gemm_stress.txt
timeout 300 python tools/attn_stress.py 200 > $O/attn_stress.txt 2>&1; echo "attn_stress rc=$?" >> $O/attn_stress.txt; tail -3 $O/attn_stress.txt
