#!/bin/bash
# round 5, run Q: do the vision tower's 64-row tails pay on the 128-tile (split-K) path instead of the skinny kernel? bench switch 6000 + m
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_q
export MJV_LIBRARY=$GRAFT_REPO_ROOT/mj-video_amd/libmjv_hip_bench.so
for rnd in 1 2; do
for codes in "" "--gemm-code 6063" "--gemm-code 6063 --gemm-code 4001"; do
  echo "== $codes" >> gpurun_out/r05_q/ab.txt
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-latency --no-secondary $codes 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = p['kernels']
print(p['value'], p['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if n.startswith(('gemm128', 'gemm64'))})" >> gpurun_out/r05_q/ab.txt
done
done
cat gpurun_out/r05_q/ab.txt
