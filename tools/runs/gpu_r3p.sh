R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -c 1500 $O/bench_default.json
