R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ac; mkdir -p $O
cd $R
timeout 300 python tools/var_check.py 5 3 2>&1 | grep -v amdgpu.ids | tee $O/var_check_5.txt
cat > /tmp/only.py <<'PY'
import sys
s=open('tools/gemm_bench.py').read()
s=s.replace('if os.environ.get("MJV_BENCH_TAILS"):','shapes=[x for x in shapes if x[0] in ("vit_qkv","vit_fc1_nogelu","llm_wqkv","square4k")]\nif os.environ.get("MJV_BENCH_TAILS"):')
open('/tmp/gemm_bench_sel.py','w').write(s.replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))',repr(sys.argv[1])))
PY
python /tmp/only.py $R
MJV_BENCH_ROUNDS=6 timeout 600 python /tmp/gemm_bench_sel.py 1000 1009 1005 7001 2>/dev/null | tee $O/direct_store_ab.txt
