R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ab; mkdir -p $O
cd $R
timeout 300 ./tools/micro/store_pattern | tee $O/store_pattern.txt
