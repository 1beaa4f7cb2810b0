R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
MJV_BENCH_TAILS=1 MJV_BENCH_ROUNDS=4 timeout 600 python tools/gemm_bench.py 5000 > $O/tails.log 2>&1; grep -v amdgpu.ids $O/tails.log
python bench.py --no-latency > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2w/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:20]: print(f"{k:26s} {v['ms_per_step']:7.3f} {v.get('tflops')}")
PY
