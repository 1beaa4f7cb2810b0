#!/usr/bin/env python3
"""Diagnostic: is the round-3 attention kernel (choice 0) further from the reference than the round-2 one (choice 5)?
(a) |hip - ref| and |hip - fp32| score rms over the C1 / C2 rank sets under each choice; (b) the attention kernel alone
against an fp64 softmax(QK^T)V on ViT- and LLM-shaped random inputs."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_e2e_gpu as T  # noqa: E402
from mj_video_amd import ops  # noqa: E402

cuda = torch.device("cuda", 0)

# (b) kernel alone
torch.manual_seed(0)
for (H, D, L, nseq, causal, scale_in) in ((16, 64, 1025, 8, False, 1.0), (16, 64, 1025, 8, False, 3.0), (16, 128, 2186, 2, True, 1.0), (16, 128, 2186, 2, True, 3.0)):
    q = (torch.randn(nseq * L, H, D, device=cuda) * scale_in).to(torch.bfloat16)
    k = (torch.randn(nseq * L, H, D, device=cuda) * scale_in).to(torch.bfloat16)
    v = torch.randn(nseq * L, H, D, device=cuda).to(torch.bfloat16)
    cu = torch.arange(0, nseq + 1, device=cuda, dtype=torch.int32) * L
    qd, kd, vd = (t.double().view(nseq, L, H, D).transpose(1, 2) for t in (q, k, v))
    s = qd @ kd.transpose(-1, -2) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.ones(L, L, device=cuda, dtype=torch.bool).triu(1), float("-inf"))
    ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(nseq * L, H, D)
    for variant in (0, 5):
        ops.attention_set_variant(variant)
        o = torch.empty(nseq * L, H * D, device=cuda, dtype=torch.bfloat16)
        ops.attention(q.view(nseq * L, H * D), k.view(nseq * L, H * D), v.view(nseq * L, H * D), o, cu, L, H, 1, D, causal,
                      D ** -0.5, 0)
        e = (o.view(nseq * L, H, D).double() - ref)
        print(f"attention D={D} L={L} causal={causal} input x{scale_in}: choice {variant}  rms err {e.pow(2).mean().sqrt().item():.3e}  "
              f"max {e.abs().max().item():.3e}  rel rms {(e.pow(2).mean() / ref.pow(2).mean()).sqrt().item():.3e}")
    del s, ref, qd, kd, vd

# (a) rank sets
for name, ppf in (("rankset_c2", 4), ("rankset_c1", 8)):
    npz, meta = T.load_golden(name)
    ref = npz["ref_bf16"][..., 0]
    f32 = npz["ref_fp32"][..., 0]
    have = ~np.isnan(f32[:, 0])
    print(name, "pairs", ref.shape[0], "fp32 pairs", int(have.sum()),
          "reference |bf16 - fp32| rms", float(np.sqrt(((ref[have] - f32[have]) ** 2).mean())))
    for variant in (0, 5):
        ops.attention_set_variant(variant)
        T._RANK_CACHE.clear()
        got = T._rank_run(cuda, name, ppf)["got"][..., 0]
        d = got - ref
        d32 = got[have] - f32[have]
        print(f"  attention choice {variant}: |hip - ref| rms {np.sqrt((d ** 2).mean()):.4f} max {np.abs(d).max():.4f}   "
              f"|hip - fp32| rms {np.sqrt((d32 ** 2).mean()):.4f} max {np.abs(d32).max():.4f}")
