R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4am; mkdir -p $O
cd $R
timeout 1300 python tools/fuzz_kernels.py 1000 7 > $O/fuzz_1000s_seed7.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_1000s_seed7.txt; tail -3 $O/fuzz_1000s_seed7.txt
