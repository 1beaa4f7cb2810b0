R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
export MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_bench.so
for rep in 1 2; do
for code in 1000 1009; do
timeout 600 python bench.py --no-cpu-baseline --no-latency --gemm-code $code > $O/bench_$code.json 2> $O/bench_$code.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r3h/bench_$code.json').read().strip().splitlines()[-1])
k=d['kernels']
print($code, d['value'], d['ms_per_step'], 'bias', k['gemm256_bias']['ms_per_step'], 'silu', k['gemm256_silu_mul']['ms_per_step'], 'gelu', k['gemm256_bias_gelu']['ms_per_step'], 'scale_res', k['gemm256_scale_res']['ms_per_step'])
PY
done; done
