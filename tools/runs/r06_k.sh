#!/bin/bash
# round 6, run K: two HIP streams (two half-batches side by side) against the one-stream step
set -x
mkdir -p gpurun_out/r06_k
timeout 900 python tools/two_stream_ab.py 6 2 > gpurun_out/r06_k/two_stream_ab.txt 2>&1
tail -12 gpurun_out/r06_k/two_stream_ab.txt
