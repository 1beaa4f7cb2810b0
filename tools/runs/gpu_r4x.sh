R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4x; mkdir -p $O
cd $R
MJV_ATTN_MODE2=1 timeout 600 python tools/attn_bench.py 20 4 0,8 2>/dev/null | grep -v vit | tee $O/attn_k8.txt

