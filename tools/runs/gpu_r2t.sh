R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t; mkdir -p $O
cd $R
timeout 600 python tools/gemm_stamps.py > $O/stamps.log 2>&1; cat $O/stamps.log | tail -30
