R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gelu or folded or race_screen or relu" > $O/pytest_gelu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gelu.log
tail -5 $O/pytest_gelu.log
# A/B on one box: the GELU GEMM of this tree and of the previous commit
cat > /tmp/onlyfc1.py <<'PY'
import re,sys
s=open('tools/gemm_bench.py').read()
s=s.replace('if os.environ.get("MJV_BENCH_TAILS"):','shapes=[x for x in shapes if x[0].startswith("vit_fc1")]\nif os.environ.get("MJV_BENCH_TAILS"):')
open('/tmp/gemm_bench_fc1.py','w').write(s.replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))',repr(sys.argv[1])))
PY
python /tmp/onlyfc1.py $R
for i in 1 2; do
MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_bench.so MJV_BENCH_ROUNDS=5 timeout 300 python /tmp/gemm_bench_fc1.py 256 2>/dev/null | sed 's/^/new  /' | tee -a $O/gelu_ab.txt
MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_bench_prev.so MJV_BENCH_ROUNDS=5 timeout 300 python /tmp/gemm_bench_fc1.py 256 2>/dev/null | sed 's/^/prev /' | tee -a $O/gelu_ab.txt
done
