R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O
cd $R
bash tools/collect_profiles.sh r02d > $O/collect.log 2>&1; tail -3 $O/collect.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $O/pmc_attn -o p -- python3 $R/tools/attn_bench.py 3 1 > $O/pmc_attn.log 2>&1; echo "pmc attn rc=$?"
