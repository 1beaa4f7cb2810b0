R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O
cd $R
timeout 300 python tools/attn_bench.py 20 3 0,1,2,3 > $O/attn_variants.txt 2>&1; echo "attn rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_calib -o p -- $R/tools/micro/issue_model calib > $O/pmc_calib.log 2>&1; echo "calib rc=$?"
cat $O/attn_variants.txt
