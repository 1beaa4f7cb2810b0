#!/bin/bash
mkdir -p gpurun_out/r3i
timeout 900 python tools/c3_diag.py > gpurun_out/r3i/c3_diag.log 2>&1
echo "diag rc=$?"
tail -12 gpurun_out/r3i/c3_diag.log
