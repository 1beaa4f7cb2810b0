R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
cat > /tmp/onlyfc1.py <<'PY'
import re,sys
s=open('tools/gemm_bench.py').read()
s=s.replace('if os.environ.get("MJV_BENCH_TAILS"):','shapes=[x for x in shapes if x[0].startswith("vit_fc1")]\nif os.environ.get("MJV_BENCH_TAILS"):')
open('/tmp/gemm_bench_fc1.py','w').write(s.replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))',repr(sys.argv[1])))
PY
python /tmp/onlyfc1.py $R
for W in 0.02 0.05; do echo "wstd $W" | tee -a $O/gelu_ab.txt
for L in bench bench_prev bench_probe1 bench_probe2; do
MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_$L.so MJV_BENCH_WSTD=$W MJV_BENCH_ROUNDS=5 timeout 300 python /tmp/gemm_bench_fc1.py 256 2>/dev/null | sed "s/^/$L  /" | tee -a $O/gelu_ab.txt
done
done
