R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4y; mkdir -p $O
cd $R
timeout 1200 python tools/fp8_attn_side_study.py > $O/fp8_attn_side_study.txt 2> $O/err.txt; echo "rc=$?" >> $O/fp8_attn_side_study.txt
cat $O/fp8_attn_side_study.txt; tail -5 $O/err.txt
