R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ad; mkdir -p $O
cd $R
timeout 900 python tools/attn_bench.py 50 3 0,8 x > $O/attn_bench.txt 2>&1; grep -E "vit_d64|d64_L1024" $O/attn_bench.txt
