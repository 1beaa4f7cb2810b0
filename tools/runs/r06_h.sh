#!/bin/bash
# round 6, run H: vision tower in chunks of 16 / 32 tiles against all 64 at once (bench A/B); 300 s of the extended fuzzer (head_dim 96,
# tile 2); tile-loop stamps of the three attention kernels in the production numerics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_h
for c in 0 32 16 0 32; do
  if [ $c = 0 ]; then python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-secondary > gpurun_out/r06_h/bench_chunk_$c.json 2>/dev/null
  else MJV_BENCH_VIT_CHUNK=$c python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-secondary > gpurun_out/r06_h/bench_chunk_$c.json 2>/dev/null; fi
  python - <<PY
import json
p = json.loads(open('gpurun_out/r06_h/bench_chunk_$c.json').read().strip().splitlines()[-1])
k = p['kernels']
print('chunk', $c, 'value', p['value'], 'ms', p['ms_per_step'], {t: k[t]['ms_per_step'] for t in ('gemm256_bias', 'gemm256_bias_gelu', 'gemm256_scale_res', 'attn_d64', 'layernorm', 'gemm64_scale_res', 'gemm64_bias')})
PY
done | tee gpurun_out/r06_h/vit_chunk_ab.txt
timeout 400 python tools/fuzz_kernels.py 300 20261005 > gpurun_out/r06_h/fuzz_300s.txt 2>&1; tail -4 gpurun_out/r06_h/fuzz_300s.txt | cut -c1-300
python tools/attn_stamps.py flash > gpurun_out/r06_h/attn_stamps_flash.txt 2>&1; grep -v amdgpu gpurun_out/r06_h/attn_stamps_flash.txt
