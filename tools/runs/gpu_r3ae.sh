R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ae; mkdir -p $O
cd $R
timeout 600 python tools/step_bubble.py > $O/bubble.txt 2>&1; grep -v amdgpu.ids $O/bubble.txt
