#!/bin/bash
# round 5, run R: the skinny kernel with eight K-tiles in flight: GEMM tests (all tile kernels), race screens, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_r
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm or sweep or rope or heads or folded" 2>&1 | tail -4 > gpurun_out/r05_r/pytest_gemm.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = p['kernels']
print(p['value'], p['ms_per_step'], p['latency']['one_video_per_forward_ms'], {n: v['ms_per_step'] for n, v in k.items() if n.startswith('gemm64')})" >> gpurun_out/r05_r/bench.txt
done
python tools/single_video_profile.py 2>/dev/null | grep "gemm64\|sum" >> gpurun_out/r05_r/bench.txt
cat gpurun_out/r05_r/pytest_gemm.txt gpurun_out/r05_r/bench.txt
