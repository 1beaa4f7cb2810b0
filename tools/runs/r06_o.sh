#!/bin/bash
# round 6, run O: rocprofv3 evidence for BASELINE configs[4] as its own line (bench.py --backbone 4b): kernel statistics and the
# two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) -> profiles/r06_4b_pmc_traffic.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06_o
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1 -o k1 -- python3 $R/bench.py --backbone 4b --steps 5 --warmup 2 > $O/bench_4b_under_rocprof.json 2> $O/k1.log
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --backbone 4b --steps 2 --warmup 1 --no-prof > /dev/null 2> $O/fetch.log
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --backbone 4b --steps 2 --warmup 1 --no-prof > /dev/null 2> $O/write.log
python3 $R/tools/pmc_summary.py --json $O/pmc_traffic_4b.json --workload "configs[4] workload (bench.py --backbone 4b: InternViT + Phi-3-mini, 4 pairs @448^2)" $O/fetch $O/write > $O/pmc_summary_4b.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O $O/k1 | head -30; head -12 $O/pmc_summary_4b.txt | cut -c1-160; tail -c 300 $O/k1.log
