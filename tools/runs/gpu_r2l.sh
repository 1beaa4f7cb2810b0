R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; mkdir -p $O
cd $R
timeout 600 python tools/gemm_stamps.py > $O/gemm_stamps.txt 2>&1; echo "rc=$?"; cat $O/gemm_stamps.txt | grep -v amdgpu.ids
