#!/bin/bash
# round 5, run Z: MFMA-pipe utilisation per kernel of the headline step (two PMC passes: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05_z
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/mfma.log
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm -o g -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-prof > /dev/null 2> $O/grbm.log
python3 $R/tools/mfma_busy_summary.py $O/mfma $O/grbm > $O/mfma_busy.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
cat $O/mfma_busy.txt | cut -c1-140
