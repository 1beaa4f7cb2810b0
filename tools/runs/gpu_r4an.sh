R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4an; mkdir -p $O
cd $R
timeout 1200 python tools/mfma_fp8_accumulation_tail.py 2>&1 | grep -v amdgpu.ids | tee $O/mfma_fp8_accumulation_tail.txt
