# attention PMC counters of the round-3 kernel (choice 0) and the round-2 kernel (choice 5), same groups as tools/runs/gpu_r2a.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/g$i -o p -- python3 $R/tools/attn_bench.py 3 1 0,5 > $O/pmc_g$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 tools/attn_counters.py $O/pmc > $O/attn_counters.txt; cat $O/attn_counters.txt | head -120
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
