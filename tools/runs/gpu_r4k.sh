R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
export MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_bench.so
for rep in 1 2; do
for code in 7001 7000 7002; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --gemm-code $code > $O/bench_nt_${code}_$rep.json 2>$O/err.txt
  python -c "
import json;d=json.load(open('$O/bench_nt_${code}_$rep.json'));k=d['kernels'];print('code',$code,'rep',$rep,d['value'],d['ms_per_step'],{n:k[n]['ms_per_step'] for n in ('gemm256_bias','gemm256_bias_gelu','gemm256_scale_res','gemm256_silu_mul','gemm256_rope_qkv','layernorm','rmsnorm','attn_d64')})"
done; done
unset MJV_LIBRARY
timeout 300 python bench.py --fp8 --steps 20 --warmup 3 > $O/bench_fp8.json 2>>$O/err.txt; python -c "
import json;d=json.load(open('$O/bench_fp8.json'));print('fp8',d['value'],d['ms_per_step'])"
