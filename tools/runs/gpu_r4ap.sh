R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ap; mkdir -p $O
cd $R
timeout 1900 python tools/fuzz_kernels.py 1500 31337 > $O/fuzz_1500s_seed31337.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_1500s_seed31337.txt; tail -4 $O/fuzz_1500s_seed31337.txt
