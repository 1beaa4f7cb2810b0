R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 600 python tools/gemm_stress.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -3 $O/stress.log
