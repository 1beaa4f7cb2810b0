R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 600 python tools/attn_stress.py 300 2>&1 | grep -E "repeat|differ" | tail -8
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
