R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r; mkdir -p $O
cd $R
MJV_BENCH_ROUNDS=5 timeout 900 python tools/gemm_bench.py 0 1005 1008 > $O/gb.log 2>&1; grep -v "nogelu\|amdgpu.ids" $O/gb.log
