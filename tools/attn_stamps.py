#!/usr/bin/env python3
"""s_memtime stamps of attn2_kernel's tile loop (diagnostic build: make -C mj-video_amd/csrc stamps -> libmjv_hip_stamps.so).
Prints the average cycles per key tile and per segment of the loop for the waves of ordinary query blocks; shares, not lengths
(the stamps drain the LDS queue at every boundary)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
os.environ["MJV_LIBRARY"] = os.path.join(ROOT, "mj-video_amd", "libmjv_hip_stamps.so")   # make -C mj-video_amd/csrc stamps
from mj_video_amd import _lib, ops  # noqa: E402

lib = _lib.load_library()
lib.mjv_attention_stamp_buffer.restype = C.c_int
lib.mjv_attention_stamp_buffer.argtypes = [C.c_void_p]
dev, BF = "cuda", torch.bfloat16
NAMES = ["loop overhead", "wait own DMA", "barrier", "issue next DMA", "slot0 QK(0)", "slot1 QK(1)|SM(0)",
         "slot2 PV(0),QK(2)|SM(1)", "slot3 PV(1),QK(3)|SM(2)", "slot4 PV(2)|SM(3)", "slot5 PV(3)", "-", "general-path tile"]


def run(name, n_seq, L, H, G, D, causal, mode, nw=4):
    N = n_seq * L
    q = torch.randn(N, H * D, device=dev).to(BF)
    k = torch.randn(N, (H // G) * D, device=dev).to(BF)
    v = torch.randn(N, (H // G) * D, device=dev).to(BF)
    o = torch.empty(N, H * D, device=dev, dtype=BF)
    cu = torch.arange(0, (n_seq + 1) * L, L, dtype=torch.int32, device=dev)
    nblk = 8 * ((((L + 63) // 64 + 2) * H * n_seq + 7) // 8) + 64
    buf = torch.zeros(nblk * nw * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode)
    torch.cuda.synchronize()
    assert lib.mjv_attention_stamp_buffer(buf.data_ptr()) == 0
    ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode)
    torch.cuda.synchronize()
    lib.mjv_attention_stamp_buffer(None)
    b = buf.view(-1, 16).cpu().numpy()
    for kind, label in ((2, "full waves"), (5, "query-0 block (one sub-block)")):
        rows = b[(b[:, 14] == kind) & (b[:, 12] > 0)]
        if not len(rows):
            continue
        tiles = rows[:, 12].sum()
        print(f"{name}: {label}: {len(rows)} waves, {tiles / len(rows):.1f} tiles each, {rows[:, 13].sum() / tiles:.0f} cycles per tile")
        for i, nm in enumerate(NAMES):
            if rows[:, i].sum():
                print(f"    {nm:28s} {rows[:, i].sum() / tiles:8.0f}  ({100.0 * rows[:, i].sum() / rows[:, 13].sum():5.1f} %)")


if len(sys.argv) > 1 and sys.argv[1] == "flash":     # the production numerics (score_round_mode 2) + the head_dim 96 kernel of round 6
    run("vit_d64 flash", 64, 1025, 16, 1, 64, False, 2)
    run("llm_d128c flash", 8, 2186, 16, 2, 128, True, 2)
    run("phi3_d96c flash", 8, 2184, 32, 1, 96, True, 2)
else:
    run("vit_d64", 64, 1025, 16, 1, 64, False, 0)
    run("llm_d128c", 8, 2186, 16, 2, 128, True, 1)
