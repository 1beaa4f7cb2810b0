R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4af; mkdir -p $O
cd $R
timeout 300 python tools/var_check.py 2 3 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|load_library" | tee $O/var_check_2.txt
cat > /tmp/only.py <<'PY'
import sys
s=open('tools/gemm_bench.py').read()
s=s.replace('if os.environ.get("MJV_BENCH_TAILS"):','shapes=[x for x in shapes if x[0] in ("vit_proj","vit_fc2","llm_wo","llm_w2_16384")] + [("llm_wo_16384", 16384, 2048, 2048, ops.EPI_SCALE_RES), ("vit_proj_65536", 65536, 1024, 1024, ops.EPI_SCALE_RES), ("vit_fc2_65536", 65536, 1024, 4096, ops.EPI_SCALE_RES)]\nif os.environ.get("MJV_BENCH_TAILS"):')
open('/tmp/gemm_bench_sel.py','w').write(s.replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))',repr(sys.argv[1])))
PY
python /tmp/only.py $R
MJV_BENCH_ROUNDS=6 timeout 600 python /tmp/gemm_bench_sel.py 1000 1002 2>/dev/null | tee $O/persistent_scale_res_ab.txt
