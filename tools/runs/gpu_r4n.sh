R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log; tail -4 $O/pytest_attn.log
MJV_ATTN_MODE2=1 timeout 300 python tools/attn_bench.py 30 3 > $O/attn_bench.txt 2>&1; grep -v amdgpu.ids $O/attn_bench.txt | grep -v 28810
