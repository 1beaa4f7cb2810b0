R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q -k "folded or trimming or single_layer_at or tiny_cases or rank_agreement_c2" > $O/pytest_gpu2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu2.log; grep -v "^$" $O/pytest_gpu2.log | tail -12
