# Collects the rocprofv3 evidence committed under profiles/: run on the GPU box from the repo root
#   bash tools/collect_profiles.sh <tag>        (one gpurun call; outputs under gpurun_out/prof_<tag>/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1 -o k1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-secondary > $O/bench_under_rocprof.json 2> $O/k1.log
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/fetch.log
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/write.log
python3 $R/tools/pmc_summary.py --json $O/pmc_traffic.json $O/fetch $O/write > $O/pmc_summary.txt
# the fp8 FFN weight path (bench.py --fp8: its own line, never the headline): kernel stats of the same command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k8 -o k8 -- python3 $R/bench.py --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/bench_fp8_under_rocprof.json 2> $O/k8.log
cd $R && timeout 900 python3 bench.py --fp8 > $O/bench_fp8.json 2> $O/bench_fp8.log
cd $R && timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O $O/k1 | head -30; tail -c 600 $O/bench_default.json
