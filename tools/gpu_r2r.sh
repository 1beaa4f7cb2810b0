R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r; mkdir -p $O
cd $R
timeout 600 python tools/var_check.py 5 10 > $O/check.log 2>&1; echo "check rc=$?"; tail -2 $O/check.log
MJV_BENCH_ROUNDS=5 timeout 900 python tools/gemm_bench.py 0 1005 > $O/gb.log 2>&1; grep -v "nogelu\|amdgpu.ids" $O/gb.log
