R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
MJV_BENCH_ROUNDS=6 timeout 900 python tools/gemm_bench.py 0 2003 2004 2005 2006 2007 2008 2010 2012 > $O/gb.log 2>&1; grep -v "^square\|nogelu\|amdgpu.ids" $O/gb.log | sed 's/ ms / /g; s/ TF\/s//g; s/tile20//g'
