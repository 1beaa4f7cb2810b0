R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3af; mkdir -p $O
cd $R
timeout 600 python tools/step_bubble.py > $O/bubble.txt 2>&1; grep -v amdgpu.ids $O/bubble.txt
timeout 2400 python -m pytest tests/test_e2e_gpu.py tests/test_preprocess_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
