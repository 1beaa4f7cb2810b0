#!/usr/bin/env python3
"""A/B of the 128 x 256 two-workgroups-per-CU GEMM (tile code 2, round 6) against the automatic choice (256 x 256 one-per-CU
kernel + its tail launches) on the short-K Linears of the vision tower and, for the record, the language tower's shapes:
bit-equality of the outputs first (same k order of the fp32 sums), then HIP-event timings in rotated rounds, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mj_video_amd import ops
from mj_video_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_SCALE_RES
dev, BF = "cuda", torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SHAPES = [("vit_qkv", 65600, 3072, 1024, EPI_BIAS), ("vit_fc1", 65600, 4096, 1024, EPI_BIAS_GELU), ("vit_proj", 65600, 1024, 1024, EPI_SCALE_RES),
          ("vit_fc2", 65600, 1024, 4096, EPI_SCALE_RES), ("llm_wo", 16976, 2048, 2048, EPI_SCALE_RES), ("llm_w2", 16976, 2048, 8192, EPI_SCALE_RES),
          ("phi_qkv", 17472, 9216, 3072, EPI_BIAS), ("phi_o", 17472, 3072, 3072, EPI_SCALE_RES), ("phi_down", 17472, 3072, 8192, EPI_SCALE_RES),
          ("mlp1_1", 16384, 2048, 4096, EPI_BIAS_GELU)]
if len(sys.argv) > 3:
    SHAPES = [s for s in SHAPES if s[0] in sys.argv[3].split(",")]
ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
ops.set_gemm_workspace(ws)
g = torch.Generator(device=dev).manual_seed(1)
for name, M, N, K, epi in SHAPES:
    a = (torch.randn(M, K, device=dev, generator=g) * 1.0).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).to(BF)
    bias = (torch.randn(N, device=dev, generator=g) * 0.1).to(BF)
    scale = (1 + 0.1 * torch.randn(N, device=dev, generator=g)).to(BF) if epi == EPI_SCALE_RES and "vit" in name else None
    res0 = torch.randn(M, N, device=dev, generator=g).to(BF) if epi == EPI_SCALE_RES else None

    def run(tile, out, res):
        ops.gemm(a, w, out, epi, bias=bias if "llm" not in name and "phi" not in name else None, scale=scale, res=res, tile=tile)

    outs = {}
    for tile in (0, 2):
        out = res0.clone() if res0 is not None else torch.full((M, N), float("nan"), dtype=BF, device=dev)
        run(tile, out, out if res0 is not None else None)     # in place on the residual stream, as the model calls it
        outs[tile] = out
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[2])
    nbad = int((outs[0] != outs[2]).sum()) if not same else 0
    fin = bool(torch.isfinite(outs[2].float()).all())
    res = {0: [], 2: []}
    buf = res0.clone() if res0 is not None else torch.empty(M, N, dtype=BF, device=dev)
    for rnd in range(rounds):
        for tile in ((0, 2) if rnd % 2 == 0 else (2, 0)):
            for _ in range(3):
                run(tile, buf, buf if res0 is not None else None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(tile, buf, buf if res0 is not None else None)
            e1.record(); torch.cuda.synchronize()
            res[tile].append(e0.elapsed_time(e1) / iters)
    fl = 2.0 * M * N * K
    t0, t2 = sorted(res[0])[len(res[0]) // 2], sorted(res[2])[len(res[2]) // 2]
    print(f"{name:9s} {M:6d} x {N:5d} x {K:5d}  auto {t0 * 1e3:8.1f} us {fl / t0 / 1e9:7.1f} TF/s | tile 2 {t2 * 1e3:8.1f} us {fl / t2 / 1e9:7.1f} TF/s "
          f"({(t0 / t2 - 1) * 100:+5.1f} %)  outputs {'bit-identical' if same else f'DIFFER in {nbad} elements'}{'' if fin else ' NON-FINITE'}", flush=True)
