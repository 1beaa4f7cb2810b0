# round-2 first GPU call: sanity of the tree, issue-model microbenchmark, default bench line, attention PMC counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
timeout 300 tools/micro/issue_model > $O/issue_model.txt 2>&1; echo "issue_model rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_attn/g$i -o p -- python3 $R/tools/attn_bench.py 3 1 > $O/pmc_attn_g$i.log 2>&1
  echo "pmc group $i rc=$?"
done
find $O -name "*kernel_trace.csv" -size +5M -delete
tail -c 1500 $O/pytest.log; cat $O/issue_model.txt | head -100; tail -c 1200 $O/bench_default.json
