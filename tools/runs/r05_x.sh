#!/bin/bash
# round 5, run X: the two PMC passes again after a comment-only edit of gemm_fp8.hip (the traffic file names the sources' hash)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05_x
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/fetch.log
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/write.log
python3 $R/tools/pmc_summary.py --json $O/pmc_traffic.json $O/fetch $O/write > $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
head -5 $O/pmc_summary.txt; head -3 $O/pmc_traffic.json
