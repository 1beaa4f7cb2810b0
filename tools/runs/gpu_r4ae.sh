R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ae; mkdir -p $O
cd $R
MJV_TEST_ATTENTION_SCORES=eager timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x > $O/pytest_e2e_eager.log 2>&1; echo "eager rc=$?" >> $O/pytest_e2e_eager.log; tail -3 $O/pytest_e2e_eager.log
MJV_TEST_NORM_FUSION=1 timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q > $O/pytest_e2e_fused.log 2>&1; echo "fused rc=$?" >> $O/pytest_e2e_fused.log; tail -8 $O/pytest_e2e_fused.log
