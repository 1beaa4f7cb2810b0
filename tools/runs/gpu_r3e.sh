R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout 600 python tools/attn_stress.py 400 2>&1 | tail -8
