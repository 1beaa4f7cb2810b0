#!/usr/bin/env python3
"""How long the host needs to ENQUEUE one forward (Python + ctypes + HIP launches) against how long the GPU needs to run
it: decides whether hipGraph capture would buy anything (it does when enqueue time approaches GPU time)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mj_video_amd import configuration as C, synth
from mj_video_amd.modeling import InternVLChatRewardModeling
from mj_video_amd.chat_input import num_image_tokens_per_tile
import bench
dev = torch.device("cuda:0")
for (S, F, pairs) in [(448, 8, 4), (224, 8, 4), (224, 8, 1)]:
    cfg = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(S), **C.mjvideo_head_kwargs())
    model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
    bench.random_init_on_device(model, cfg, dev, seed=1)
    model.config.pad_token_id = synth.PAD_ID; model.model.img_context_token_id = synth.IMG_CONTEXT_ID; model.eval()
    n = 2 * pairs
    px = torch.randn(n * F, 3, S, S, device=dev).to(torch.bfloat16)
    ids, mask = synth.pad_batch([synth.synth_input_ids(num_image_tokens_per_tile(cfg) * F, caption_seed=p // 2) for p in range(n)])
    ids, mask = ids.to(dev), mask.to(dev)
    for _ in range(3): model.forward(px, ids, mask)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.forward(px, ids, mask); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"{S}px x{F} frames, {pairs} pairs: host enqueue {1e3 * min(enq):.1f} ms, forward complete {1e3 * min(tot):.1f} ms")
    del model
    torch.cuda.empty_cache()
