"""Thin tensor-level wrappers over the C ABI: torch supplies device memory and the current HIP stream,
every computation happens in libmjv_hip.so.  Every function enqueues on the current HIP stream OF THE DEVICE ITS
TENSORS LIVE ON (``torch.cuda.current_stream(t.device)``) and makes that device the current one for the call (the
library reads the current device for its CU count and per-device kernel attributes), so a direct ``ops.*`` call on
cuda:1 tensors works whatever the caller's current device is.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_ROPE_QKV, EPI_SCALE_RES, EPI_SILU_MUL, FMT_MXFP8, AttnDesc,
                   GemmDesc, HeadsDesc, check, load_library)

BF16 = torch.bfloat16


def _stream(t: torch.Tensor) -> int:
    """HIP stream handle for work on ``t``: the current stream of t's device (a model on cuda:1 must not be launched
    on cuda:0's stream just because cuda:0 is the current device)."""
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_bf16(*ts):
    for t in ts:
        if t is not None and (t.dtype != BF16 or not t.is_cuda):
            raise TypeError(f"expected a bf16 device tensor, got {t.dtype} on {t.device}")


def _row_stride(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "rows must be contiguous"
    return t.stride(0)


class MX8:
    """An MXFP8 matrix on the GPU (include/mjv.h "MXFP8 operand format"): ``data`` uint8 [rows, cols] OCP e4m3 elements, ``scales``
    uint8 e8m0 block scales in the MFMA's lane layout (``mxfp8_scale_bytes(rows, cols)`` bytes).  Operand / output of the fp8
    FFN GEMMs; produced by ``quantize_mxfp8``, the ``*_mxfp8`` norms and ``gemm(..., out=MX8)``."""
    __slots__ = ("data", "scales")

    def __init__(self, data: torch.Tensor, scales: torch.Tensor):
        assert data.dtype == torch.uint8 and scales.dtype == torch.uint8 and data.is_cuda and scales.is_cuda
        assert data.dim() == 2 and data.stride(1) == 1 and data.shape[1] % 128 == 0, "MXFP8: [rows, cols % 128 == 0] row-contiguous"
        assert scales.is_contiguous() and scales.numel() >= mxfp8_scale_bytes(data.shape[0], data.shape[1])
        self.data, self.scales = data, scales

    @staticmethod
    def empty(rows: int, cols: int, device) -> "MX8":
        return MX8(torch.empty(rows, cols, dtype=torch.uint8, device=device),
                   torch.empty(mxfp8_scale_bytes(rows, cols), dtype=torch.uint8, device=device))

    @property
    def shape(self):
        return self.data.shape

    @property
    def device(self):
        return self.data.device

    def rows(self, n: int) -> "MX8":
        """the first ``n`` rows (same scale buffer: the records are indexed by absolute row; the GROUP count of the scale
        layout follows the row count, so only a view with the same ceil(rows / 64) addresses the same records)"""
        assert (n + 63) // 64 == (self.data.shape[0] + 63) // 64, "a row view must keep the number of 64-row scale groups"
        return MX8(self.data[:n], self.scales)


def mxfp8_scale_bytes(rows: int, cols: int) -> int:
    return (cols // 128) * ((rows + 63) // 64) * 256


def quantize_mxfp8(x: torch.Tensor, out: Optional[MX8] = None) -> MX8:
    """bf16 [rows, cols] -> MXFP8 (weights once at load; tests)."""
    _chk_bf16(x)
    rows, cols = x.shape
    out = MX8.empty(rows, cols, x.device) if out is None else out
    assert tuple(out.shape) == (rows, cols)
    with torch.cuda.device(x.device):
        check(load_library().mjv_quantize_mxfp8(x.data_ptr(), _row_stride(x), out.data.data_ptr(), _row_stride(out.data),
                                                out.scales.data_ptr(), rows, cols, _stream(x)), "mjv_quantize_mxfp8")
    return out


class _CallDefaults(threading.local):
    """Per-THREAD defaults of the per-call choices the C ABI takes in its descriptors (ABI 4 has no process-wide setter):
    the split-K scratch of this thread's GEMM calls, the forced tile kernel and the attention kernel (parity tests).  Two
    threads scoring on two streams each see their own."""
    gemm_ws: Optional[torch.Tensor] = None
    tile: int = 0
    attn_kernel: int = 0


_tls = _CallDefaults()


def gemm_workspace_bytes() -> int:
    return int(load_library().mjv_gemm_workspace_bytes())


def set_gemm_workspace(ws: Optional[torch.Tensor]) -> None:
    """Scratch buffer handed to every following ``gemm`` call of THIS THREAD that does not pass its own ``workspace``
    (uint8/any dtype, on the GPU, private to the stream the calls go to while they are in flight).  None: never split K."""
    if ws is not None:
        assert ws.is_cuda and ws.is_contiguous() and ws.data_ptr() % 16 == 0
    _tls.gemm_ws = ws


def _gemm_mxfp8(a: MX8, w: MX8, out, epilogue: int, bias, scale, res, M: Optional[int], workspace=None, tile=None):
    """MXFP8 operands (include/mjv.h, ABI 5): ``out`` a bf16 tensor or an ``MX8`` (the epilogue's result block-quantised)."""
    assert isinstance(w, MX8), "MXFP8 activations need MXFP8 weights"
    _chk_bf16(bias, scale, res)
    d = GemmDesc()
    d.A, d.lda = a.data.data_ptr(), _row_stride(a.data)
    d.W, d.ldw = w.data.data_ptr(), _row_stride(w.data)
    d.M = a.shape[0] if M is None else M
    assert d.M == a.shape[0], "an MXFP8 operand's scale records are laid out for its own row count"
    d.N, d.K = w.shape[0], w.shape[1]
    assert a.shape[1] == d.K, (a.shape, w.shape)
    d.a_format = d.w_format = FMT_MXFP8
    d.a_scales, d.w_scales = a.scales.data_ptr(), w.scales.data_ptr()
    nout = d.N // 2 if epilogue == EPI_SILU_MUL else d.N
    if isinstance(out, MX8):
        assert tuple(out.shape) == (d.M, nout), (tuple(out.shape), d.M, nout)
        d.C, d.ldc, d.c_format, d.c_scales = out.data.data_ptr(), _row_stride(out.data), FMT_MXFP8, out.scales.data_ptr()
        dev = out.data.device
    else:
        _chk_bf16(out)
        d.C, d.ldc = out.data_ptr(), _row_stride(out)
        dev = out.device
    d.epilogue = epilogue
    d.bias, d.scale = _p(bias), _p(scale)
    d.res, d.ldr = _p(res), (_row_stride(res) if res is not None else 0)
    # (the split-K scratch: peeled tail rows / under-filled launches run K-sliced; a forced tile = the one unsliced launch)
    ws = workspace if workspace is not None else _tls.gemm_ws
    if ws is not None and ws.device == dev:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    d.tile = _tls.tile if tile is None else tile
    if d.tile not in (0, 256):
        d.tile = 256    # (one tile kernel exists for MXFP8 operands: the parity tests' tile sweep maps onto it)
    with torch.cuda.device(dev):
        check(load_library().mjv_gemm_bf16(C.byref(d), torch.cuda.current_stream(dev).cuda_stream), "mjv_gemm_bf16(mxfp8)")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_BIAS,
         bias: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, res_mod: int = 0, res_off: int = 0, out_group: int = 0,
         out_pad: int = 0, out_rows: Optional[torch.Tensor] = None, M: Optional[int] = None,
         rope: Optional[tuple] = None, workspace: Optional[torch.Tensor] = None, tile: Optional[int] = None,
         folded_norm: Optional[tuple] = None) -> torch.Tensor:
    """out = epilogue(a[M,K] @ w[N,K]^T); 2-D row-contiguous bf16 views (row strides are honoured).
    ``rope`` (EPI_ROPE_QKV only) = (cos, sin, positions, q_out, k_out, group): see include/mjv.h.
    ``workspace``: split-K scratch of THIS call (default: the calling thread's ``set_gemm_workspace`` buffer);
    ``tile``: 0 automatic, 64 / 128 / 256 force one tile kernel (default: the calling thread's ``gemm_set_tile`` value).
    ``a`` and ``w`` may be ``MX8`` (MXFP8 operands, ABI 5), ``out`` then a bf16 tensor or an ``MX8``.
    ``folded_norm`` = (row_scale,) or (row_scale, row_shift, col_shift, bias_f32): fp32 vectors of a norm folded into this GEMM
    (include/mjv.h "row_scale"; row vectors of ``padded_rows(M)`` entries from ``row_stats``, column vectors of
    ``padded_rows(N)``)."""
    if isinstance(a, MX8):
        assert res_mod == 0 and out_group == 0 and out_rows is None and rope is None, "MXFP8 GEMMs write plain rows"
        return _gemm_mxfp8(a, w, out, epilogue, bias, scale, res, M, workspace, tile)
    _chk_bf16(a, w, out, bias, scale, res)
    lib = load_library()
    d = GemmDesc()
    d.A, d.lda = a.data_ptr(), _row_stride(a)
    d.W, d.ldw = w.data_ptr(), _row_stride(w)
    d.C, d.ldc = out.data_ptr(), _row_stride(out)
    d.M = a.shape[0] if M is None else M
    d.N, d.K = w.shape[0], w.shape[1]
    assert a.shape[1] == d.K, (a.shape, w.shape)
    d.epilogue = epilogue
    d.bias, d.scale = _p(bias), _p(scale)
    d.res, d.ldr = _p(res), (_row_stride(res) if res is not None else 0)
    d.res_mod, d.res_off, d.out_group, d.out_pad = res_mod, res_off, out_group, out_pad
    if out_rows is not None:
        assert out_rows.dtype == torch.int32 and out_rows.is_cuda
    d.out_rows = _p(out_rows)
    if epilogue == EPI_ROPE_QKV:
        cos, sin, positions, q_out, k_out, group = rope
        _chk_bf16(cos, sin, q_out, k_out)
        assert positions.dtype == torch.int32 and positions.is_cuda and positions.numel() >= d.M
        assert cos.shape[1] == 128 and cos.is_contiguous() and sin.is_contiguous()
        d.rope_cos, d.rope_sin, d.rope_pos = cos.data_ptr(), sin.data_ptr(), positions.data_ptr()
        d.rope_q, d.rope_k = _p(q_out), k_out.data_ptr()   # (q_out None with group 0: a k | v projection, no q heads)
        d.rope_ldq, d.rope_ldk, d.rope_group = (_row_stride(q_out) if q_out is not None else 0), _row_stride(k_out), group
    if folded_norm is not None:
        for t, n in zip(folded_norm, (d.M, d.M, d.N, d.N)):
            assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous() and t.numel() >= padded_rows(n), (t.shape, n)
        d.row_scale = folded_norm[0].data_ptr()
        if len(folded_norm) == 4:
            d.row_shift, d.col_shift, d.bias_f32 = (t.data_ptr() for t in folded_norm[1:])
        else:
            assert len(folded_norm) == 1
    ws = workspace if workspace is not None else _tls.gemm_ws
    if ws is not None and ws.device == a.device:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    d.tile = _tls.tile if tile is None else tile
    with torch.cuda.device(out.device):   # the library plans against / sets attributes on the CURRENT device
        check(lib.mjv_gemm_bf16(C.byref(d), _stream(out)), "mjv_gemm_bf16")
    return out


def gemm_set_tile(tile: int) -> None:
    """This THREAD's default tile kernel: 0 = automatic choice, 64 / 128 / 256 = force that kernel (parity tests).  Codes
    >= 1000 are the measurement switches of the bench build (include/mjv_bench.h; process-wide) and need that library
    (MJV_LIBRARY=.../libmjv_hip_bench.so)."""
    if tile in (0, 64, 128, 256):
        _tls.tile = tile
        if tile == 0 and hasattr(load_library(), "mjv_bench_gemm_set") and getattr(load_library().mjv_bench_gemm_set, "argtypes", None):
            check(load_library().mjv_bench_gemm_set(1000), "mjv_bench_gemm_set")   # "automatic" also leaves the kernel variants
        return
    lib = load_library()
    if getattr(getattr(lib, "mjv_bench_gemm_set", None), "argtypes", None) is None:
        raise _lib.MjvLibraryError(f"gemm_set_tile({tile}): measurement switches exist in the bench build only "
                                   "(make -C mj-video_amd/csrc bench; MJV_LIBRARY=<repo>/mj-video_amd/libmjv_hip_bench.so)")
    check(lib.mjv_bench_gemm_set(tile), "mjv_bench_gemm_set")


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, cu_seqlens: torch.Tensor,
              max_seqlen: int, n_heads: int, kv_group: int, head_dim: int, causal: bool, scale: float,
              score_round_mode: int, q_head_stride: Optional[int] = None, k_head_stride: Optional[int] = None,
              v_head_stride: Optional[int] = None, o_head_stride: Optional[int] = None,
              kernel: Optional[int] = None, cu_seqlens_q: Optional[torch.Tensor] = None, max_seqlen_q: int = 0,
              prefix_k: Optional[torch.Tensor] = None, prefix_v: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q/k/v/out are 2-D views [rows, >= heads*head_dim] (row stride honoured, first head at column 0).
    ``kernel``: 0 automatic, 4 / 5 = the round-1 / round-2 kernels (default: the calling thread's ``attention_set_variant``).
    Causal only (include/mjv.h ABI 6): ``cu_seqlens_q`` / ``max_seqlen_q`` - the queries are the LAST rows of every sequence,
    packed at those offsets of q / out; ``prefix_k`` / ``prefix_v`` [P, ...] - P (a multiple of 64) shared keys / values in front of
    every sequence's own rows, same row stride and head stride as k / v."""
    _chk_bf16(q, k, v, out)
    assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_cuda
    lib = load_library()
    d = AttnDesc()
    d.Q, d.K, d.V, d.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = _row_stride(q), _row_stride(k), _row_stride(v), _row_stride(out)
    d.q_head_stride = head_dim if q_head_stride is None else q_head_stride
    d.k_head_stride = head_dim if k_head_stride is None else k_head_stride
    d.v_head_stride = head_dim if v_head_stride is None else v_head_stride
    d.o_head_stride = head_dim if o_head_stride is None else o_head_stride
    d.cu_seqlens = cu_seqlens.data_ptr()
    d.n_seqs, d.max_seqlen = cu_seqlens.numel() - 1, max_seqlen
    d.n_heads, d.kv_group, d.head_dim = n_heads, kv_group, head_dim
    d.causal, d.scale, d.score_round_mode = int(causal), scale, score_round_mode
    d.kernel = _tls.attn_kernel if kernel is None else kernel
    if cu_seqlens_q is not None:
        assert cu_seqlens_q.dtype == torch.int32 and cu_seqlens_q.is_cuda and cu_seqlens_q.numel() == cu_seqlens.numel()
        d.cu_seqlens_q, d.max_seqlen_q = cu_seqlens_q.data_ptr(), max_seqlen_q
    if prefix_k is not None or prefix_v is not None:
        _chk_bf16(prefix_k, prefix_v)
        assert _row_stride(prefix_k) == d.ldk and _row_stride(prefix_v) == d.ldv and prefix_k.shape[0] == prefix_v.shape[0]
        d.prefix_k, d.prefix_v, d.prefix_len = prefix_k.data_ptr(), prefix_v.data_ptr(), prefix_k.shape[0]
    with torch.cuda.device(out.device):
        check(lib.mjv_attention_bf16(C.byref(d), _stream(out)), "mjv_attention_bf16")
    return out


def attention_set_variant(v: int) -> None:
    """This THREAD's default attention kernel: 0 automatic, 4 = register-staged round-1 kernel, 5 = round-2 choice, 6 / 7 =
    round-3 kernel with two / four waves per workgroup (all correct; tests A/B them).  1-3 are measurement variants of the
    bench build (process-wide, bench library only)."""
    lib = load_library()
    bench = getattr(getattr(lib, "mjv_bench_attention_set", None), "argtypes", None) is not None
    if v in (0, 4, 5, 6, 7):
        _tls.attn_kernel = v
        if bench:
            check(lib.mjv_bench_attention_set(0), "mjv_bench_attention_set")
        return
    if not bench:
        raise _lib.MjvLibraryError(f"attention_set_variant({v}): measurement variants exist in the bench build only")
    _tls.attn_kernel = 5 if v in (1, 2, 3) else 0
    check(lib.mjv_bench_attention_set(v), "mjv_bench_attention_set")


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, eps: float,
              rows: Optional[int] = None, gather_grid: int = 0) -> torch.Tensor:
    _chk_bf16(x, gamma, beta, out)
    lib = load_library()
    rows = out.shape[0] if rows is None else rows
    with torch.cuda.device(out.device):
        check(lib.mjv_layernorm_bf16(x.data_ptr(), _row_stride(x), out.data_ptr(), _row_stride(out), gamma.data_ptr(),
                                     beta.data_ptr(), rows, out.shape[1], eps, gather_grid, _stream(out)),
              "mjv_layernorm_bf16")
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, eps: float,
            row_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk_bf16(x, w, out)
    lib = load_library()
    if row_index is not None:
        assert row_index.dtype == torch.int32 and row_index.is_cuda
    with torch.cuda.device(out.device):
        check(lib.mjv_rmsnorm_bf16(x.data_ptr(), _row_stride(x), out.data_ptr(), _row_stride(out), w.data_ptr(),
                                   _p(row_index), out.shape[0], out.shape[1], eps, _stream(out)), "mjv_rmsnorm_bf16")
    return out


def layernorm_mxfp8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: MX8, eps: float) -> MX8:
    """``layernorm`` with the bf16 result block-quantised on the way out (the input of the fp8 fc1 GEMM)."""
    _chk_bf16(x, gamma, beta)
    assert tuple(out.shape) == tuple(x.shape)
    with torch.cuda.device(x.device):
        check(load_library().mjv_layernorm_mxfp8(x.data_ptr(), _row_stride(x), out.data.data_ptr(), _row_stride(out.data),
                                                 out.scales.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.shape[0], x.shape[1],
                                                 eps, _stream(x)), "mjv_layernorm_mxfp8")
    return out


def rmsnorm_mxfp8(x: torch.Tensor, w: torch.Tensor, out: MX8, eps: float) -> MX8:
    _chk_bf16(x, w)
    assert tuple(out.shape) == tuple(x.shape)
    with torch.cuda.device(x.device):
        check(load_library().mjv_rmsnorm_mxfp8(x.data_ptr(), _row_stride(x), out.data.data_ptr(), _row_stride(out.data),
                                               out.scales.data_ptr(), w.data_ptr(), x.shape[0], x.shape[1], eps, _stream(x)),
              "mjv_rmsnorm_mxfp8")
    return out


def padded_rows(n: int) -> int:
    """entries a row / column vector of a folded norm must hold for n rows / columns (whole 256-tiles are fetched)"""
    return (n + 255) // 256 * 256


def row_stats(x: torch.Tensor, rstd: torch.Tensor, mean_rstd: Optional[torch.Tensor], eps: float) -> None:
    """rstd[m] (and mean[m] * rstd[m]: LayerNorm; None: RMSNorm statistics) of the rows of ``x`` - the producer half of a norm
    folded into its consuming GEMM (``gemm(..., folded_norm=...)``)."""
    _chk_bf16(x)
    assert rstd.dtype == torch.float32 and rstd.is_cuda and rstd.numel() >= x.shape[0]
    assert mean_rstd is None or (mean_rstd.dtype == torch.float32 and mean_rstd.numel() >= x.shape[0])
    with torch.cuda.device(x.device):
        check(load_library().mjv_row_stats_bf16(x.data_ptr(), _row_stride(x), rstd.data_ptr(), _p(mean_rstd), x.shape[0], x.shape[1],
                                                eps, _stream(x)), "mjv_row_stats_bf16")


def rope_split(qkv: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
               positions: torch.Tensor, kv_heads: int, group: int) -> None:
    _chk_bf16(qkv, q, k, cos, sin)
    assert positions.dtype == torch.int32 and positions.is_cuda
    assert cos.shape[1] == 128 and cos.is_contiguous() and sin.is_contiguous()
    lib = load_library()
    with torch.cuda.device(qkv.device):
        check(lib.mjv_rope_split_bf16(qkv.data_ptr(), _row_stride(qkv), q.data_ptr(), _row_stride(q), k.data_ptr(),
                                      _row_stride(k), cos.data_ptr(), sin.data_ptr(), positions.data_ptr(),
                                      qkv.shape[0], kv_heads, group, _stream(qkv)), "mjv_rope_split_bf16")


def rope_heads(x: torch.Tensor, n_heads: int, head_stride: int, cos: torch.Tensor, sin: torch.Tensor,
               positions: torch.Tensor) -> None:
    """Rotary embedding IN PLACE on the first ``n_heads`` heads (``head_stride`` elements apart, from column 0) of every row of
    ``x``; ``cos`` / ``sin`` [positions, rot_dim] bf16 tables (include/mjv.h, ABI 7: the Phi-3 tower's [q | k | v] projection)."""
    _chk_bf16(x, cos, sin)
    assert positions.dtype == torch.int32 and positions.is_cuda and positions.numel() >= x.shape[0]
    assert cos.is_contiguous() and sin.is_contiguous() and cos.shape == sin.shape
    assert n_heads * head_stride <= x.shape[1]
    with torch.cuda.device(x.device):
        check(load_library().mjv_rope_heads_bf16(x.data_ptr(), _row_stride(x), head_stride, n_heads, cos.shape[1], cos.data_ptr(),
                                                 sin.data_ptr(), positions.data_ptr(), x.shape[0], _stream(x)), "mjv_rope_heads_bf16")


def patchify(pixels: torch.Tensor, patches: torch.Tensor, patch: int) -> torch.Tensor:
    _chk_bf16(pixels, patches)
    assert pixels.is_contiguous() and pixels.dim() == 4 and pixels.shape[1] == 3 and pixels.shape[2] == pixels.shape[3]
    lib = load_library()
    with torch.cuda.device(patches.device):
        check(lib.mjv_patchify_bf16(pixels.data_ptr(), patches.data_ptr(), _row_stride(patches), pixels.shape[0],
                                    pixels.shape[2], patch, _stream(patches)), "mjv_patchify_bf16")
    return patches


def cls_rows(x: torch.Tensor, cls: torch.Tensor, pos0: torch.Tensor, tiles: int, tokens_per_tile: int) -> None:
    _chk_bf16(x, cls, pos0)
    lib = load_library()
    with torch.cuda.device(x.device):
        check(lib.mjv_cls_rows_bf16(x.data_ptr(), _row_stride(x), cls.data_ptr(), pos0.data_ptr(), tiles,
                                    tokens_per_tile, x.shape[1], _stream(x)), "mjv_cls_rows_bf16")


def embed_gather(ids: torch.Tensor, table: torch.Tensor, x: torch.Tensor, skip_id: int) -> None:
    _chk_bf16(table, x)
    assert ids.dtype == torch.int32 and ids.is_cuda
    lib = load_library()
    with torch.cuda.device(x.device):
        check(lib.mjv_embed_gather_bf16(ids.data_ptr(), table.data_ptr(), _row_stride(table), x.data_ptr(),
                                        _row_stride(x), ids.numel(), x.shape[1], skip_id, table.shape[0], _stream(x)),
              "mjv_embed_gather_bf16")


def reward_heads(desc: HeadsDesc, device: torch.device) -> None:
    """``device``: where the descriptor's pointers live (a descriptor carries no tensor to take the stream from)."""
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        check(load_library().mjv_reward_heads_bf16(C.byref(desc), stream), "mjv_reward_heads_bf16")


# ------------------------------------------------------------------------------------ profiler access
def prof_enable(on: bool) -> None:
    check(load_library().mjv_prof_enable(int(on)), "mjv_prof_enable")


def prof_filter(tag: Optional[str]) -> None:
    """Record only launches tagged ``tag`` (None: every kernel)."""
    check(load_library().mjv_prof_filter(tag.encode() if tag else None), "mjv_prof_filter")


def prof_reset() -> None:
    check(load_library().mjv_prof_reset(), "mjv_prof_reset")


def prof_results() -> dict:
    """{tag: dict(launches, ms, flops, bytes)} after synchronising the recorded events."""
    lib = load_library()
    check(lib.mjv_prof_collect(), "mjv_prof_collect")
    out = {}
    for i in range(lib.mjv_prof_count()):
        name, n = C.c_char_p(), C.c_int64()
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        check(lib.mjv_prof_get(i, C.byref(name), C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "mjv_prof_get")
        out[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
    return out
