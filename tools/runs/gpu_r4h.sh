R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "folded or row_stats" > $O/pytest_folded.log 2>&1; echo "pytest rc=$?" >> $O/pytest_folded.log; tail -5 $O/pytest_folded.log
for nf in 0 1; do
  if [ $nf = 1 ]; then export MJV_TEST_NORM_FUSION=1; else unset MJV_TEST_NORM_FUSION; fi
  timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "single_layer_at_production or full_c1 or full_c2 or engineered_c1 or tiny_cases" > $O/gate_nf$nf.log 2>&1; echo "rc=$?" >> $O/gate_nf$nf.log
  echo "== norm_fusion=$nf"; grep -E "HIP vs reference|engineered|preference agreement|spearman|rms\(|passed|failed|rc=" $O/gate_nf$nf.log | head -40
done
unset MJV_TEST_NORM_FUSION
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_nf0.json 2>$O/bench_nf0.err
MJV_BENCH_NORM_FUSION=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_nf1.json 2>$O/bench_nf1.err
python - <<'PY'
import json
for nf in (0, 1):
    try:
        d = json.load(open(f"gpurun_out/r4h/bench_nf{nf}.json"))
        k = d["kernels"]
        norms = {n: v["ms_per_step"] for n, v in k.items() if "norm" in n or "row_stats" in n}
        print("norm_fusion", nf, d["value"], "pairs/s", d["ms_per_step"], "ms;", norms, "gemm ms:", round(sum(v["ms_per_step"] for n, v in k.items() if n.startswith("gemm")), 2))
    except Exception as e:
        print(nf, "failed", e)
PY
