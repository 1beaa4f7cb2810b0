#!/bin/bash
# round 5, run W: after the fp8 GELU epilogue port - the whole GPU suite, then the profile collection on the (new) final kernel sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_w
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-300 | tail -30 > gpurun_out/r05_w/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_w/smoke.txt 2>&1
bash tools/collect_profiles.sh r05_w > gpurun_out/collect_r05_w.log 2>&1
grep -n "FAILED\|passed\|failed" gpurun_out/r05_w/pytest.txt; tail -2 gpurun_out/r05_w/smoke.txt | cut -c1-120; tail -c 300 gpurun_out/collect_r05_w.log
