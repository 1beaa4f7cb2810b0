R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/collect_profiles.sh r02c > $O/collect.log 2>&1; tail -5 $O/collect.log
