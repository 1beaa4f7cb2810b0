R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout 600 python tools/norm_ab.py > $O/norm_ab.txt 2>&1; grep -v amdgpu.ids $O/norm_ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "persistent or gemm_bias or silu" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
