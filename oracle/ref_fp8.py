"""ORACLE - test infrastructure, never the product path.

MXFP8 operand rounding for the fp8 FFN weight path (BASELINE configs[4]) as plain numpy / PyTorch CPU code:
the DEFINITION the HIP kernels (mj-video_amd/csrc/mx8.h, mxfp8.hip, gemm_fp8.hip, the *_mxfp8 norms) are held to.

Parity status: PARITY UNPINNED against the reference - the reference has no fp8 path at all (its README names the 2B bf16
checkpoint only; the 4B backbone of BASELINE configs[4] has no implementation in /root/reference, see
scripts/model/internvl2/modeling_internvl_chat.py:125-130).  What IS pinned: the bf16 model around the five FFN Linears is
oracle/ref_cpu.py (bit-identical to the reference); the operand format follows the OCP Microscaling Formats (MX) v1.0
specification, "MXFP8" with E4M3 elements, 32-element blocks along the reduction dimension and an E8M0 shared scale, as
consumed by gfx950's v_mfma_scale_f32_16x16x128_f8f6f4.  One choice differs from the spec's example conversion: the scale is
the SMALLEST power of two that brings the block's largest magnitude inside the e4m3 range (<= 448), so no element saturates
(the spec's floor(log2(amax)) - 8 lets amax in (448, 512) * 2^e clamp).

Only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s ``cpu_baseline`` and ``tools/`` may import this file.
"""
from __future__ import annotations

import contextlib
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_cpu

BLOCK = 32
E4M3_MAX = 448.0


# --------------------------------------------------------------------------- OCP e4m3fn, by table
def _e4m3_decode_table() -> np.ndarray:
    """value of every non-negative finite code 0x00 .. 0x7e (S.EEEE.MMM, bias 7, subnormals, no infinities; 0x7f = NaN)"""
    v = np.zeros(127, dtype=np.float64)
    for c in range(127):
        e, m = c >> 3, c & 7
        v[c] = m * 2.0 ** -9 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 7)
    return v


_VALS = _e4m3_decode_table()
_MIDS = (_VALS[:-1] + _VALS[1:]) / 2.0


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float array -> uint8 e4m3fn codes, round to nearest even; |x| must be <= 448 (the quantiser guarantees it)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    if a.size and float(a.max()) > E4M3_MAX:
        raise ValueError("e4m3_encode: magnitude beyond 448 (the block scale should have prevented this)")
    idx = np.searchsorted(_MIDS, a, side="left")                 # number of midpoints strictly below a
    tie = (idx < _MIDS.size) & (a == _MIDS[np.minimum(idx, _MIDS.size - 1)])
    idx = np.where(tie & (idx % 2 == 1), idx + 1, idx)            # a tie goes to the even code
    return (idx.astype(np.uint8) | np.where(np.signbit(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def e4m3_decode(c: np.ndarray) -> np.ndarray:
    c = np.asarray(c, dtype=np.uint8)
    mag = _VALS[np.minimum(c & 0x7F, 126)]
    return np.where(c & 0x80, -mag, mag).astype(np.float32)


# --------------------------------------------------------------------------- block quantiser
def mx_quantize(x: torch.Tensor) -> Tuple[np.ndarray, np.ndarray]:
    """bf16-representable values [rows, K] (K % 32 == 0) -> (codes uint8 [rows, K], scale bytes uint8 [rows, K / 32]).

    Per 32-element block: amax as a bf16 bit pattern u = E:8 | m:7;  b = max(1, E - 8 + (m > 0x60)) - the smallest power of
    two 2^(b-127) with amax / 2^(b-127) <= 448, kept >= 2^-126 so the scale is a normal fp32 -; element = e4m3_rne(x / 2^(b-127)).
    """
    xb = x.detach().to(torch.bfloat16).contiguous()
    rows, K = xb.shape
    assert K % BLOCK == 0
    u = (xb.view(torch.int16).numpy().astype(np.int32) & 0x7FFF).reshape(rows, K // BLOCK, BLOCK)
    amax = u.max(axis=2)
    b = np.maximum(((amax + 0x1F) >> 7) - 8, 1).astype(np.int32)
    xf = xb.float().numpy().astype(np.float64).reshape(rows, K // BLOCK, BLOCK)
    scaled = xf * np.exp2((127 - b).astype(np.float64))[:, :, None]
    return e4m3_encode(scaled).reshape(rows, K), b.astype(np.uint8)


def mx_dequantize(codes: np.ndarray, sb: np.ndarray) -> torch.Tensor:
    rows, K = codes.shape
    v = e4m3_decode(codes).astype(np.float64).reshape(rows, K // BLOCK, BLOCK) * np.exp2(sb.astype(np.float64) - 127.0)[:, :, None]
    return torch.from_numpy(v.reshape(rows, K).astype(np.float32))   # (exact: <= 4 significant bits times a power of two)


def mx_fake_quant(x: torch.Tensor) -> torch.Tensor:
    """the fp32 values an MXFP8 GEMM multiplies: dequantize(quantize(x)), any leading shape"""
    shp = x.shape
    q, sb = mx_quantize(x.reshape(-1, shp[-1]))
    return mx_dequantize(q, sb).reshape(shp)


def mx_fake_quant_fast(x: torch.Tensor) -> torch.Tensor:
    """``mx_fake_quant`` in multi-threaded torch ops, for the 2B-dims checks (1.4 G weight elements): the same block scale
    from the same bit arithmetic, the element rounding by torch's float8_e4m3fn cast - equal to the table encoder above on
    every bf16 value inside the format's range and on random scaled data (tests/test_oracle_golden.py pins the two to each
    other); exact power-of-two scaling both ways."""
    shp = x.shape
    xb = x.detach().to(torch.bfloat16).reshape(-1, shp[-1] // BLOCK, BLOCK)
    u = xb.view(torch.int16).to(torch.int32) & 0x7FFF
    b = (((u.amax(dim=2) + 0x1F) >> 7) - 8).clamp_min(1)                      # [rows, blocks]
    xf = xb.float()
    scaled = torch.ldexp(xf, (127 - b)[:, :, None].expand_as(xf))
    q = scaled.to(torch.float8_e4m3fn).float()
    return torch.ldexp(q, (b - 127)[:, :, None].expand_as(q)).reshape(shp)


def mx_scale_records(sb: np.ndarray) -> np.ndarray:
    """scale bytes [rows, K / 32] -> the byte image of include/mjv.h's scale layout (K % 128 == 0): per K-tile of 128 and
    64-row group one 256-byte record, byte (row % 16) * 16 + kb * 4 + (row / 16) % 4.  Bytes of rows beyond ``rows`` are 0."""
    rows, nb = sb.shape
    assert nb % 4 == 0
    groups = (rows + 63) // 64
    out = np.zeros((nb // 4, groups, 256), dtype=np.uint8)
    r = np.arange(rows)
    for blk in range(nb):
        out[blk // 4, r >> 6, (r & 15) * 16 + (blk & 3) * 4 + ((r >> 4) & 3)] = sb[:, blk]
    return out.reshape(-1)


def mx_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fp8 FFN Linear: both operands rounded to MXFP8 along K, products and sums in fp32 (exact products; the summation
    order is the GEMM's own - compared within an accumulation-order tolerance), + bias, ONE rounding to the activation dtype -
    the bf16 Linear's rounding point (modeling_intern_vit.py:256-257, modeling_internlm2.py:256-258)."""
    y = F.linear(mx_fake_quant(x), mx_fake_quant(w))
    if b is not None:
        y = y + b.float()
    return y.to(x.dtype)


@contextlib.contextmanager
def fp8_ffn(fast: bool = True):
    """inside: oracle/ref_cpu.py's five FFN Linears (fc1, fc2, w1, w3, w2) round both operands to MXFP8 (``mx_linear``).
    ``fast``: the torch-op form of the rounding (pinned to the table form by the CPU tests); a weight's rounded copy is NOT
    kept between calls (a 2B-dims forward would hold 5.6 GB of fp32 copies): every Linear re-rounds its weight."""
    prev = ref_cpu._ffn_linear
    fq = mx_fake_quant_fast if fast else mx_fake_quant

    def lin(x, w, b=None):
        y = F.linear(fq(x), fq(w))
        if b is not None:
            y = y + b.float()
        return y.to(x.dtype)

    ref_cpu._ffn_linear = lin
    try:
        yield
    finally:
        ref_cpu._ffn_linear = prev


def reward_forward_fp8(*args, fast: bool = True, **kwargs):
    """oracle/ref_cpu.reward_forward with the fp8 FFN path"""
    with fp8_ffn(fast):
        return ref_cpu.reward_forward(*args, **kwargs)
