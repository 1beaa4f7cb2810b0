#!/bin/bash
# round 6, run R: the compiler's scheduling strategy (-mllvm --amdgpu-sched-strategy=max-ilp / max-memory-clause) on gemm.hip and on
# attention.hip: four variant libraries (one object replaced each, built in the container for this run) against the product library,
# the 4-pair step, interleaved three times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_r
P=$GRAFT_REPO_ROOT/mj-video_amd
for round in 1 2 3; do
  for v in prod gemm_ilp gemm_mem attn_ilp attn_mem; do
    lib=$P/libmjv_hip_var_$v.so; [ $v = prod ] && lib=$P/libmjv_hip.so
    MJV_LIBRARY=$lib python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-latency > gpurun_out/r06_r/bench_${v}_$round.json 2> gpurun_out/r06_r/err_${v}_$round.txt
  done
done
python - <<'PY' | tee gpurun_out/r06_r/sched_strategy_ab.txt
import json, glob, collections
res = collections.defaultdict(list); ker = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r06_r/bench_*.json')):
    v = f.split('bench_')[1].rsplit('_', 1)[0]
    try:
        p = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    res[v].append(p['ms_per_step'])
    for k, x in p['kernels'].items():
        if k.startswith(('gemm256_', 'attn_')):
            ker[k][v].append(x['ms_per_step'])
print("compiler scheduling strategy per source file: ms per 4-pair step, three interleaved runs each")
for v in ('prod', 'gemm_ilp', 'gemm_mem', 'attn_ilp', 'attn_mem'):
    print(f"{v:9s} {' '.join(f'{x:7.2f}' for x in res[v])}   best {min(res[v]):7.2f}")
print("per kernel, ms per step in the profiled step (best of three):")
for k in ker:
    print(f"  {k:22s} " + "  ".join(f"{v} {min(ker[k][v]):6.2f}" for v in ('prod', 'gemm_ilp', 'gemm_mem', 'attn_ilp', 'attn_mem') if ker[k][v]))
PY
