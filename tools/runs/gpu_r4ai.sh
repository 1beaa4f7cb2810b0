R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ai; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python3 $R/tools/preprocess_bench.py > $O/out.txt 2> $O/err.txt
find $O -name "*kernel_trace.csv" -delete
head -8 $O/k/k_kernel_stats.csv | cut -c1-200
