R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4aj; mkdir -p $O
cd $R
timeout 900 python tools/pcie_inclusive.py 2>/dev/null | tee $O/pcie_inclusive.txt
