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


# ------------------------------------------------------------------------------------------------------------------
# Pillow's 8-bit bicubic resampler, restated (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
# ImagingResampleHorizontal_8bpc / Vertical_8bpc).  ``pil_resample_coeffs`` feeds the GPU kernel; ``pil_resize_u8`` is the
# numpy statement of the same arithmetic, bit-identical to ``Image.resize`` (tests/test_host_logic.py).
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size: int, out_size: int):
    """(bounds int32 [out, 2] = (first tap, tap count), coefficients int32 [out, ksize]) of one resampling pass."""
    import math
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << _PRECISION_BITS), 0.5 + kk * (1 << _PRECISION_BITS))
    return bounds, fixed.astype(np.int64).astype(np.int32)   # C's (int) cast truncates toward zero, like astype


def pil_resize_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``np.asarray(Image.fromarray(img).resize((out_w, out_h)))`` for an H x W x 3 uint8 array, in integer numpy."""
    out = img
    half = 1 << (_PRECISION_BITS - 1)
    if out_w != out.shape[1]:
        b, k = pil_resample_coeffs(out.shape[1], out_w)
        tmp = np.empty((out.shape[0], out_w, 3), dtype=np.uint8)
        for x in range(out_w):
            x0, n = b[x]
            acc = half + (out[:, x0:x0 + n, :].astype(np.int64) * k[x, :n][None, :, None]).sum(axis=1)
            tmp[:, x, :] = np.clip(acc >> _PRECISION_BITS, 0, 255)
        out = tmp
    if out_h != out.shape[0]:
        b, k = pil_resample_coeffs(out.shape[0], out_h)
        tmp = np.empty((out_h, out.shape[1], 3), dtype=np.uint8)
        for y in range(out_h):
            y0, n = b[y]
            acc = half + (out[y0:y0 + n].astype(np.int64) * k[y, :n][:, None, None]).sum(axis=0)
            tmp[y] = np.clip(acc >> _PRECISION_BITS, 0, 255)
        out = tmp
    return out


_COEFF_CACHE = {}


def load_frames_device(frames_u8: torch.Tensor, input_size=448, max_num=1):
    """GPU version of ``load_frames`` for a stack of equally sized decoded frames.

    ``frames_u8``: uint8 ``[F, H, W, 3]`` (RGB) on the GPU (or on the host: it is copied once).  Returns
    ``(pixel_values bf16 [sum tiles, 3, S, S] on the GPU, num_patches_list)`` - bit-identical to
    ``load_frames(...)[0].to(torch.bfloat16)``: Pillow-exact integer resize, fp32 normalisation, one bf16 rounding.
    The tile grid / thumbnail rule is ``dynamic_preprocess``'s (data.py:81-117)."""
    import ctypes as C
    from . import _lib
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise TypeError(f"frames must be uint8 [F, H, W, 3], got {frames_u8.dtype} {tuple(frames_u8.shape)}")
    if not torch.cuda.is_available():
        raise RuntimeError("load_frames_device runs on the MI355X only (use load_frames on the host)")
    lib = _lib.load_library()
    dev = frames_u8.device if frames_u8.is_cuda else torch.device("cuda", torch.cuda.current_device())
    frames_u8 = frames_u8.to(dev).contiguous()
    F, H, W, _ = frames_u8.shape
    S = input_size
    cols, rows = target_grid(W, H, 1, max_num, S)
    n_grid = cols * rows
    per_frame = n_grid + (1 if n_grid != 1 else 0)   # thumbnail iff more than one tile
    out = torch.empty(F * per_frame, 3, S, S, dtype=torch.bfloat16, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    mean = (C.c_float * 3)(*IMAGENET_MEAN)
    std = (C.c_float * 3)(*IMAGENET_STD)

    def tables(n_in, n_out):
        key = (n_in, n_out, str(dev))
        if key not in _COEFF_CACHE:
            b, k = pil_resample_coeffs(n_in, n_out)
            _COEFF_CACHE[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), k.shape[1])
        return _COEFF_CACHE[key]

    def run(out_w, out_h, tile_offset):
        xb, xk, kx = tables(W, out_w)
        yb, yk, ky = tables(H, out_h)
        tmp = torch.empty(F, H, out_w, 3, dtype=torch.uint8, device=dev)
        _lib.check(lib.mjv_resize_normalize_u8(frames_u8.data_ptr(), F, H, W, out_w, out_h, xb.data_ptr(), xk.data_ptr(), kx,
                                               yb.data_ptr(), yk.data_ptr(), ky, tmp.data_ptr(), out.data_ptr(), S, per_frame,
                                               tile_offset, mean, std, stream), "mjv_resize_normalize_u8")

    run(S * cols, S * rows, 0)
    if per_frame != n_grid:
        run(S, S, n_grid)
    return out, [per_frame] * F


def decode_frames(video_path, bound=None, num_segments=32) -> np.ndarray:
    """The decode half of data.py:158-179: ``num_segments`` uniformly sampled RGB frames as one uint8 array [F, H, W, 3].
    URLs are not fetched (no network in this build): pass a local path."""
    if str(video_path).startswith("http"):
        raise RuntimeError("this build has no network access: download the video and pass a local path")
    try:
        from decord import VideoReader, cpu
    except ImportError as e:  # same hard dependency as the reference (data.py:4)
        raise ImportError("decoding video files needs the 'decord' package; "
                          "use load_frames() / load_frames_device() with already-decoded frames otherwise") from e
    vr = VideoReader(str(Path(video_path)), ctx=cpu(0), num_threads=1)
    max_frame = len(vr) - 1
    fps = float(vr.get_avg_fps())
    idx = get_index(bound, fps, max_frame, first_idx=0, num_segments=num_segments)
    return np.stack([vr[int(i)].asnumpy() for i in idx])


def load_video(video_path, bound=None, input_size=448, max_num=1, num_segments=32):
    """Drop-in for data.py:158-179 (host path: PIL resize + normalise)."""
    frames = decode_frames(video_path, bound=bound, num_segments=num_segments)
    return load_frames(list(frames), input_size=input_size, max_num=max_num)
