R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3am; mkdir -p $O
cd $R
timeout 600 python tools/silu_exhaustive.py > $O/silu.txt 2>&1; grep -v amdgpu.ids $O/silu.txt
