R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
tail -3 $O/pytest_attn.log
timeout 300 python tools/attn_bench.py 20 4 0,5 > $O/attn_bench.txt 2>&1
grep -v amdgpu $O/attn_bench.txt
timeout 300 python tools/attn_stamps.py > $O/attn_stamps.txt 2>&1
grep -v amdgpu $O/attn_stamps.txt
