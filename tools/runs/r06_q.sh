#!/bin/bash
# round 6, run Q: cache policy of the GEMM operand DMA (global_load_lds aux immediate): default against nt on the A stream, on the W
# stream, on both - four builds of the same sources, the 4-pair step, interleaved three times.  The three variant libraries were built
# for this run from gemm.hip with the builtin's last argument in stage_half() replaced by `WHICH < 2 ? MJV_AUX_W : MJV_AUX_A`
# (-DMJV_AUX_A=2 / -DMJV_AUX_W=2; 148 / 212 / 360 of the file's 678 global_load_lds_dwordx4 then carry `nt`) and linked with the
# product objects of the other files; the edit was not kept (all three are slower), so this script documents the run, it does not rebuild it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_q
P=$GRAFT_REPO_ROOT/mj-video_amd
for round in 1 2 3; do
  for v in prod A2 W2 AW2; do
    lib=$P/libmjv_hip_aux_$v.so; [ $v = prod ] && lib=$P/libmjv_hip.so
    MJV_LIBRARY=$lib python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-latency > gpurun_out/r06_q/bench_${v}_$round.json 2> gpurun_out/r06_q/err_${v}_$round.txt
  done
done
python - <<'PY' | tee gpurun_out/r06_q/aux_ab.txt
import json, glob, collections
res = collections.defaultdict(list); ker = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r06_q/bench_*.json')):
    v = f.split('bench_')[1].rsplit('_', 1)[0]
    try:
        p = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    res[v].append(p['ms_per_step'])
    for k, x in p['kernels'].items():
        if k.startswith('gemm256_'):
            ker[k][v].append(x['ms_per_step'])
print("GEMM operand DMA cache policy (aux of global_load_lds_dwordx4): ms per 4-pair step, three interleaved runs each")
for v in ('prod', 'A2', 'W2', 'AW2'):
    print(f"{v:5s} {' '.join(f'{x:7.2f}' for x in res[v])}   best {min(res[v]):7.2f}")
print("per kernel, ms per step in the profiled step (best of three):")
for k in ker:
    print(f"  {k:22s} " + "  ".join(f"{v} {min(ker[k][v]):6.2f}" for v in ('prod', 'A2', 'W2', 'AW2') if ker[k][v]))
PY
