R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "prefetch or eval_driver or mjbench or race_screen or full_c1 or full_c2" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.log; tail -5 $O/pytest_sel.log
timeout 300 python tools/pcie_inclusive.py > $O/pcie_inclusive.txt 2>&1; grep -v amdgpu.ids $O/pcie_inclusive.txt
timeout 600 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-latency > $O/bench_200_steps.json 2>$O/err.txt; python -c "
import json;d=json.load(open('$O/bench_200_steps.json'));print('200 steps:',d['value'],d['ms_per_step'])"
timeout 900 python tools/cpu_baseline_sweep.py 16 32 64 128 > $O/cpu_baseline_sweep.txt 2>&1; cat $O/cpu_baseline_sweep.txt
