R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4aq; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_fp8_gpu.py -m gpu -q -x -k "study_flag or deterministic" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
