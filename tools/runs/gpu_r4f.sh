R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mode2" > $O/pytest_mode2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_mode2.log; tail -5 $O/pytest_mode2.log
MJV_ATTN_MODE2=1 timeout 300 python tools/attn_bench.py 30 3 > $O/attn_bench_mode2.txt 2>&1; cat $O/attn_bench_mode2.txt | grep -v amdgpu.ids
for mode in eager flash; do
  MJV_TEST_ATTENTION_SCORES=$mode timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "single_layer_at_production or full_c1 or full_c2 or engineered_c1" > $O/gate_$mode.log 2>&1; echo "rc=$?" >> $O/gate_$mode.log
  grep -E "HIP vs reference|engineered|preference agreement|spearman|rms|passed|failed|rc=" $O/gate_$mode.log | head -40
done
