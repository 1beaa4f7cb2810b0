R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ao; mkdir -p $O
cd $R
timeout 900 python tools/pcie_inclusive.py > $O/pcie.txt 2>&1; grep -v amdgpu.ids $O/pcie.txt
