"""GPU frame preprocessing against the host path: Pillow resize -> tiles -> ToTensor/Normalize ->
bf16.  Byte/integer work up to the normalisation, so the bar is BIT-EXACT."""
import numpy as np
import pytest
import torch

from mj_video_amd import video

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,max_num,expect_tiles", [
    (720, 1280, 1, 1),      # the eval driver's setting: max_num=1, one 448^2 tile per frame (eval_genai_mjvideo.py:130)
    (480, 854, 1, 1),
    (448, 448, 1, 1),       # identity resize
    (300, 200, 1, 1),       # up-scaling
    (720, 1280, 6, 3),      # 2x1 grid + thumbnail
    (896, 1344, 6, 7),      # 3x2 grid + thumbnail (BASELINE.json configs[3] shape)
])
def test_device_preprocessing_is_bit_exact(cuda, H, W, max_num, expect_tiles):
    rng = np.random.default_rng(H * 7 + W)
    F = 3
    # smooth content + noise, so the resize is not averaging pure noise
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    for f in range(F):
        base = 127 + 100 * np.sin(xx / (17.0 + f) + yy / 29.0)[..., None] * np.array([1.0, 0.7, -0.8])
        frames.append(np.clip(base + rng.normal(0, 20, size=(H, W, 3)), 0, 255).astype(np.uint8))
    ref, counts = video.load_frames(frames, input_size=448, max_num=max_num)
    assert counts == [expect_tiles] * F
    got, gcounts = video.load_frames_device(torch.from_numpy(np.stack(frames)).to(cuda), input_size=448, max_num=max_num)
    assert gcounts == counts and got.dtype == torch.bfloat16 and got.shape == ref.shape
    assert torch.equal(got.cpu(), ref.to(torch.bfloat16))


def test_device_preprocessing_random_geometries(cuda):
    """25 random frame sizes (16 .. 1600 pixels a side: extreme aspect ratios, down- and up-scaling, odd sizes) x max_num in
    {1, 4, 6, 12} x input_size in {448, 224, 56}: tile counts equal and every bf16 bit equal to Pillow + the host transform"""
    import random
    rng = random.Random(5)
    nrng = np.random.default_rng(5)
    for it in range(25):
        H, W = rng.choice([16, 17, 100, 333, 448, 720, 1080, rng.randint(16, 1600)]), rng.choice([16, 31, 200, 448, 854, 1280, 1920, rng.randint(16, 1600)])
        max_num = rng.choice([1, 1, 4, 6, 12])
        size = rng.choice([448, 448, 224, 56])
        if max_num > 1 and size == 448 and H * W > 1500 * 1500:
            max_num = 6
        frames = [nrng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for _ in range(2)]
        ref, counts = video.load_frames(frames, input_size=size, max_num=max_num)
        got, gcounts = video.load_frames_device(torch.from_numpy(np.stack(frames)).to(cuda), input_size=size, max_num=max_num)
        assert gcounts == counts, (H, W, max_num, size, gcounts, counts)
        assert got.shape == ref.shape, (H, W, max_num, size)
        assert torch.equal(got.cpu(), ref.to(torch.bfloat16)), (H, W, max_num, size, int((got.cpu() != ref.to(torch.bfloat16)).sum()))


def test_device_preprocessing_feeds_the_model(cuda):
    """uint8 frames -> device preprocessing -> reward forward: same scores as the host-preprocessed path (bitwise)"""
    from util import build_hip_model, make_cfg
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    cfg = make_cfg("tiny", 56)
    model = build_hip_model(cfg, synth.synth_state_dict(cfg, seed=4, dtype=torch.float32), cuda)
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, size=(120, 160, 3), dtype=np.uint8) for _ in range(4)]
    host, _ = video.load_frames(frames, input_size=56, max_num=1)
    devp, _ = video.load_frames_device(torch.from_numpy(np.stack(frames)), input_size=56, max_num=1)
    ids = synth.synth_input_ids(num_image_tokens_per_tile(cfg) * 4, 2).to(cuda)
    a = model.forward(host.to(torch.bfloat16).to(cuda), ids, None).score
    b = model.forward(devp, ids, None).score
    assert torch.equal(a, b)
