#!/usr/bin/env python3
"""Attention micro-benchmark at the model's two shapes (HIP events, random data)."""
import sys, os
os.environ.setdefault("MJV_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mj-video_amd", "libmjv_hip_bench.so"))   # bench build: make -C mj-video_amd/csrc bench
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mj_video_amd import ops
dev = "cuda"; BF = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(name, n_seq, L, H, G, D, causal, mode):
    N = n_seq * L
    q = torch.randn(N, H * D, device=dev).to(BF); k = torch.randn(N, (H // G) * D, device=dev).to(BF)
    v = torch.randn(N, (H // G) * D, device=dev).to(BF); o = torch.empty(N, H * D, device=dev, dtype=BF)
    cu = torch.arange(0, (n_seq + 1) * L, L, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * D * n_seq * H * L * L * (0.5 if causal else 1.0)
    print(f"{name:12s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s", flush=True)


MODE2 = bool(os.environ.get("MJV_ATTN_MODE2"))   # also time score_round_mode 2 (unrounded fp32 scores) next to the eager modes
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for rnd in range(rounds):
    for v in variants[rnd % len(variants):] + variants[:rnd % len(variants)]:   # rotated: the first variant of a round is favoured
        ops.attention_set_variant(v)
        print("variant", v, end="  ")
        run("vit_d64", 64, 1025, 16, 1, 64, False, 0)
        print("variant", v, end="  ")
        run("llm_d128", 8, 2186, 16, 2, 128, True, 1)
        if MODE2 and v in (0, 6, 7):
            print("variant", v, end="  "); run("vit_d64 m2", 64, 1025, 16, 1, 64, False, 2)
            print("variant", v, end="  "); run("llm_d128 m2", 8, 2186, 16, 2, 128, True, 2)
            print("variant", v, end="  "); run("d128c_28810 m1", 1, 28810, 16, 2, 128, True, 1)
            print("variant", v, end="  "); run("d128c_28810 m2", 1, 28810, 16, 2, 128, True, 2)
        if len(sys.argv) > 4:
            print("variant", v, end="  "); run("d64_L1024", 64, 1024, 16, 1, 64, False, 0)
            print("variant", v, end="  "); run("d128_L2048", 8, 2048, 16, 2, 128, False, 1)
            print("variant", v, end="  "); run("d128c_L2048", 8, 2048, 16, 2, 128, True, 1)
            print("variant", v, end="  "); run("d128c_L8192", 2, 8192, 16, 2, 128, True, 1)
            print("variant", v, end="  "); run("d128c_L28810", 1, 28810, 16, 2, 128, True, 1)
ops.attention_set_variant(0)
