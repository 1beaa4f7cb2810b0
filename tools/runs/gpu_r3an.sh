R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3an; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=" $O/pytest.log | tail -3
