R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O
cd $R
timeout 600 python tools/attn_bench.py 50 3 0 x > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt | grep -v amdgpu.ids
timeout 600 python tools/attn_stress.py 200 2>&1 | grep -E "repeat|differ" | tail -8
timeout 900 python tools/c3_diag.py > $O/c3_diag.log 2>&1; grep -v amdgpu.ids $O/c3_diag.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
