#!/bin/bash
# round 5, run L: do the language tower's K = 2048 tails (wqkv, wo: 592 rows since the prefix cache) pay as K-sliced 256 tiles now?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_l
export MJV_LIBRARY=$GRAFT_REPO_ROOT/mj-video_amd/libmjv_hip_bench.so
for rnd in 1 2; do
for codes in "" "--gemm-code 4432" "--gemm-code 4432 --gemm-code 4304" "--gemm-code 4432 --gemm-code 4304 --gemm-code 4116"; do
  echo "== $codes" >> gpurun_out/r05_l/ab.txt
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-latency --no-secondary $codes 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = p['kernels']
print(p['value'], p['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if n.startswith(('gemm128', 'gemm256s', 'rope_split', 'gemm64_scale'))})" >> gpurun_out/r05_l/ab.txt
done
done
cat gpurun_out/r05_l/ab.txt
