R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -m gpu -q -x -k "gelu or folded or race_screen or relu or epilogue" > $O/pytest_gelu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gelu.log
tail -5 $O/pytest_gelu.log
cat > /tmp/onlyfc1.py <<'PY'
import re,sys
s=open('tools/gemm_bench.py').read()
s=s.replace('if os.environ.get("MJV_BENCH_TAILS"):','shapes=[x for x in shapes if x[0].startswith("vit_fc1")]\nif os.environ.get("MJV_BENCH_TAILS"):')
open('/tmp/gemm_bench_fc1.py','w').write(s.replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))',repr(sys.argv[1])))
PY
python /tmp/onlyfc1.py $R
for W in 0.02 0.05; do echo "wstd $W" | tee -a $O/gelu_ab.txt
for L in bench bench_prev bench bench_prev; do
MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_$L.so MJV_BENCH_WSTD=$W MJV_BENCH_ROUNDS=5 timeout 300 python /tmp/gemm_bench_fc1.py 256 2>/dev/null | sed "s/^/$L  /" | tee -a $O/gelu_ab.txt
done
done
