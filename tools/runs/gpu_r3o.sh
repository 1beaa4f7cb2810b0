R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk -o c -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-prof > /dev/null 2> $O/clk.log
cd $R
python3 tools/clock_summary.py $O/clk > $O/clock_summary.txt; cat $O/clock_summary.txt | head -24
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
