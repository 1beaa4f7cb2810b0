#!/usr/bin/env python3
"""What does one rebuild of the prompt-prefix cache cost? The headline step with a warm cache against the same step with the
cache dropped before every forward (model._build_prefix: the language tower over the 64 prefix tokens alone)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from mj_video_amd import configuration as C, synth
from mj_video_amd.modeling import InternVLChatRewardModeling
dev = torch.device('cuda:0')
cfg = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(448), **C.mjvideo_head_kwargs())
m = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
bench.random_init_on_device(m, cfg, dev, seed=1)
m.config.pad_token_id = synth.PAD_ID; m.model.img_context_token_id = synth.IMG_CONTEXT_ID; m.eval()
px, ids, mask, N = bench.synthetic_batch(cfg, dev, 4, 448, 8, seed=100)
def run(cold, n=10):
    for _ in range(2): m.forward(px, ids.clone(), mask.clone())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if cold: m._prefix = None; m._prefix_misses = 0; m._prefix_last_miss = None
        m.forward(px, ids.clone(), mask.clone())
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for r in range(2):
    print(f"warm cache {run(False):.3f} ms per step; cache rebuilt in every forward {run(True):.3f} ms per step")
