"""Frame sampling / tiling / normalisation: the caller-side ``load_video`` of the reference.

Restates scripts/data_processor/data.py:56-179 with numpy + PIL only (torchvision and decord are not
required at import time):

* ``get_index``            data.py:127-137   uniform frame indices, ``np.linspace(..., endpoint=False, dtype=int)``
* ``find_closest_aspect_ratio`` / ``dynamic_preprocess``   data.py:66-117   tile grid choice, crop order, thumbnail rule
* ``build_transform``      data.py:56-64     RGB -> bicubic resize to SxS -> /255 -> ImageNet normalise
* ``load_video``           data.py:158-179   same signature and return value

Video decoding needs ``decord`` exactly like the reference; it is imported lazily.  ``load_frames`` accepts
already-decoded frames (H x W x 3 uint8 arrays or PIL images), which is what the tests use.
Parity status: frame-index rule and tile-grid choice are pinned by values captured from the reference's own
functions; decode and torchvision's bicubic are parity-unpinned (absent offline).
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_index(bound, fps, max_frame, first_idx=0, num_segments=32):
    if bound is None:
        return np.linspace(first_idx, max_frame, num_segments, endpoint=False, dtype=int)
    start_frame, end_frame = int(bound[0] * fps), int(bound[1] * fps)
    return np.linspace(start_frame, end_frame, num_segments, endpoint=False, dtype=int)


def target_grid(width: int, height: int, min_num: int, max_num: int, image_size: int) -> Tuple[int, int]:
    """(columns, rows) of the tile grid closest to the frame's aspect ratio; ties go to the larger grid only
    when the frame has more than half of that grid's pixels (data.py:66-79,86-94)."""
    aspect = width / height
    grids = sorted({(i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                    if min_num <= i * j <= max_num}, key=lambda g: g[0] * g[1])
    best, best_diff = (1, 1), float("inf")
    area = width * height
    for g in grids:
        diff = abs(aspect - g[0] / g[1])
        if diff < best_diff:
            best, best_diff = g, diff
        elif diff == best_diff and area > 0.5 * image_size * image_size * g[0] * g[1]:
            best = g
    return best


def dynamic_preprocess(image: Image.Image, min_num=1, max_num=12, image_size=448, use_thumbnail=False) -> List[Image.Image]:
    cols, rows = target_grid(image.size[0], image.size[1], min_num, max_num, image_size)
    resized = image.resize((image_size * cols, image_size * rows))
    tiles = []
    for t in range(cols * rows):
        x, y = (t % cols) * image_size, (t // cols) * image_size
        tiles.append(resized.crop((x, y, x + image_size, y + image_size)))
    if use_thumbnail and len(tiles) != 1:
        tiles.append(image.resize((image_size, image_size)))
    return tiles


def transform_tile(tile: Image.Image, input_size: int) -> torch.Tensor:
    if tile.mode != "RGB":
        tile = tile.convert("RGB")
    if tile.size != (input_size, input_size):
        tile = tile.resize((input_size, input_size), resample=Image.BICUBIC)
    a = np.asarray(tile, dtype=np.float32) / 255.0
    a = (a - np.asarray(IMAGENET_MEAN, dtype=np.float32)) / np.asarray(IMAGENET_STD, dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def load_frames(frames: Sequence, input_size=448, max_num=1) -> Tuple[torch.Tensor, List[int]]:
    """Tiles + normalises already-decoded frames: returns (pixel_values [sum tiles, 3, S, S] f32, num_patches_list)."""
    pixel_values, num_patches = [], []
    for fr in frames:
        img = fr if isinstance(fr, Image.Image) else Image.fromarray(np.asarray(fr))
        img = img.convert("RGB")
        tiles = dynamic_preprocess(img, image_size=input_size, use_thumbnail=True, max_num=max_num)
        pv = torch.stack([transform_tile(t, input_size) for t in tiles])
        num_patches.append(pv.shape[0])
        pixel_values.append(pv)
    return torch.cat(pixel_values), num_patches


def load_video(video_path, bound=None, input_size=448, max_num=1, num_segments=32):
    """Drop-in for data.py:158-179.  URLs are not fetched (no network in this build): pass a local path."""
    if str(video_path).startswith("http"):
        raise RuntimeError("this build has no network access: download the video and pass a local path")
    try:
        from decord import VideoReader, cpu
    except ImportError as e:  # same hard dependency as the reference (data.py:4)
        raise ImportError("load_video needs the 'decord' package to decode video files; "
                          "use load_frames() with already-decoded frames otherwise") from e
    vr = VideoReader(str(Path(video_path)), ctx=cpu(0), num_threads=1)
    max_frame = len(vr) - 1
    fps = float(vr.get_avg_fps())
    idx = get_index(bound, fps, max_frame, first_idx=0, num_segments=num_segments)
    return load_frames([vr[int(i)].asnumpy() for i in idx], input_size=input_size, max_num=max_num)
