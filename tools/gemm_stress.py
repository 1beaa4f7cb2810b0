#!/usr/bin/env python3
"""Race screen for the pipelined GEMM (LDS-DMA in flight across barriers): many launches on a busy chip, integer data,
every output element checked against the exact result.  Also runs two streams concurrently (the production mode)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mj_video_amd import ops
BF = torch.bfloat16; dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
shapes = [(65600, 1024, 1024), (17488, 2048, 2048), (17488, 4096, 2048), (8200, 3072, 1024), (4133, 2048, 8192), (17488, 2048, 8192)]
bad = 0
t0 = time.time()
s2 = torch.cuda.Stream()
for (M, N, K) in shapes:
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(-1, 2, (M, K), generator=g).float().to(BF).to(dev)
    w = torch.randint(-1, 2, (N, K), generator=g).float().to(BF).to(dev)
    ref = (a.float() @ w.float().t()).to(BF)
    out1 = torch.empty(M, N, dtype=BF, device=dev); out2 = torch.empty(M, N, dtype=BF, device=dev)
    torch.cuda.synchronize()   # the side stream must not start while the default stream still uses recycled memory
    s2.wait_stream(torch.cuda.current_stream())
    n = max(4, iters * 2048 * 2048 * 2048 // (M * N * K) // 4)
    for it in range(n):
        ops.gemm(a, w, out1, ops.EPI_BIAS)
        with torch.cuda.stream(s2):
            ops.gemm(a, w, out2, ops.EPI_BIAS)
        torch.cuda.synchronize()
        e1, e2 = int((out1 != ref).sum()), int((out2 != ref).sum())
        if e1 or e2:
            bad += 1
            print("MISMATCH", (M, N, K), it, e1, e2, flush=True)
    print((M, N, K), "launches", 2 * n, "ok" if not bad else "BAD", flush=True)
print("total bad launches:", bad, "time", round(time.time() - t0, 1))
