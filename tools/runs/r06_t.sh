#!/bin/bash
# round 6, run T: MFMA-pipe utilisation per kernel (two PMC passes each: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) of the headline
# step on the final sources and of BASELINE configs[4] as its own line (bench.py --backbone 4b: head_dim 96 attention, K = 3072 GEMMs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06_t
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for w in 2b 4b; do
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma_$w -o m -- python3 $R/bench.py --backbone $w --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/mfma_$w.log
  timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm_$w -o g -- python3 $R/bench.py --backbone $w --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/grbm_$w.log
  python3 $R/tools/mfma_busy_summary.py $O/mfma_$w $O/grbm_$w > $O/mfma_busy_$w.txt
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
for w in 2b 4b; do echo "== $w"; cat $O/mfma_busy_$w.txt | cut -c1-140 | head -14; done
