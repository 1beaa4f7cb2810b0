R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
tail -5 $O/pytest_attn.log
timeout 600 python tools/attn_bench.py 50 3 0 x > $O/attn_bench.txt 2>&1; grep -v amdgpu.ids $O/attn_bench.txt
timeout 600 python tools/attn_stress.py 200 2>&1 | grep -E "repeat|differ" | tail -8
