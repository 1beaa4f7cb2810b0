R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3aa; mkdir -p $O
cd $R
timeout 900 python tools/fuzz_kernels.py 60 3 > $O/fuzz1.txt 2>&1; echo "fuzz rc=$?"; grep -v amdgpu.ids $O/fuzz1.txt | tail -40
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "randomised" 2>&1 | tail -3
