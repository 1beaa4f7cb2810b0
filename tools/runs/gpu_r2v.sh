R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v; mkdir -p $O
cd $R
timeout 600 python tools/var_check.py 0 3 > $O/check.log 2>&1; echo "check rc=$?"; tail -1 $O/check.log
timeout 600 python tools/gemm_stamps.py 2>&1 | grep "per tile" | sed 's/; launch span.*//'
MJV_BENCH_ROUNDS=5 timeout 900 python tools/gemm_bench.py 0 > $O/gb.log 2>&1; grep -v "nogelu\|amdgpu.ids" $O/gb.log
