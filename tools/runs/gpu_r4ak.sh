R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ak; mkdir -p $O
cd $R
timeout 600 python tools/single_video_profile.py 2>/dev/null | tee $O/single_video_profile.txt
