R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|Error|assert|engineered|rho|agreement|layer probes|rms\(hip|c4lite|C4 layer|heads vs|shard" $O/pytest.log | tail -60
