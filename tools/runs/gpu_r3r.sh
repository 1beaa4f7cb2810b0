R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
tail -5 $O/pytest_attn.log
timeout 600 python tools/single_video_profile.py > $O/single.txt 2>&1; grep -v amdgpu.ids $O/single.txt | head -12
