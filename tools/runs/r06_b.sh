#!/bin/bash
# round 6, run B: first GPU run of the Phi-3 tower (ABI 7): head_dim 96 attention, rope_heads, tiny e2e vs the transformers-pinned
# fixtures; the attention suite of the 2B path after the swizzle refactor
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_b
python -m pytest tests/test_phi3_gpu.py -m gpu -q -x -s -k "not single_layer and not full" 2>&1 | tail -40 > gpurun_out/r06_b/pytest_phi3.txt
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention" 2>&1 | tail -8 > gpurun_out/r06_b/pytest_attn.txt
cat gpurun_out/r06_b/pytest_phi3.txt | cut -c1-300; cat gpurun_out/r06_b/pytest_attn.txt | cut -c1-300
