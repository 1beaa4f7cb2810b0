R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
timeout 600 python tools/single_video_profile.py > $O/single.txt 2>&1; grep -v amdgpu.ids $O/single.txt
