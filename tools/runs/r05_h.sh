#!/bin/bash
# round 5, run H: the whole GPU suite on the sources with the offset headroom and the prefix-only pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_h
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -120 > gpurun_out/r05_h/pytest.txt
grep -n "FAILED\|passed\|failed" gpurun_out/r05_h/pytest.txt
