"""Per-kernel parity on the MI355X: every C-ABI op against a plain PyTorch fp32 reference of the same op
with the reference's bf16 rounding points.  Tolerances are stated per test: bf16 has 8 significant bits,
so one unit in the last place is 2^-8 relative; fp32 accumulation ORDER differs between an MFMA tile
loop and a CPU GEMM, which can flip the final rounding by 1 ulp (2 ulps after a second rounding point).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import bf16_ulps

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(*shape, std=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(BF)


def assert_close_bf16(out, ref, max_ulps, frac_exact=0.0, atol=0.0, what=""):
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    ulps = bf16_ulps(out, ref)
    ok = (ulps <= max_ulps) | ((out - ref).abs() <= atol)
    assert ok.all(), (f"{what}: {int((~ok).sum())} of {ok.numel()} elements off by more than {max_ulps} bf16 ulps "
                      f"(max {int(ulps.max())}, max abs err {(out - ref).abs().max().item():.3e})")
    if frac_exact:
        fe = (ulps == 0).float().mean().item()
        assert fe >= frac_exact, f"{what}: only {fe:.4f} of elements bit-identical (< {frac_exact})"


# ------------------------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[64, 128, 256])
def tile(request, cuda):
    """run every GEMM test on all three tile kernels (the automatic choice would hide the 256^2 kernel at test sizes and
    the 64 x 32 skinny kernel above 128 rows)"""
    from mj_video_amd import ops
    ops.gemm_set_tile(request.param)
    yield request.param
    ops.gemm_set_tile(0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 128), (1, 256, 256), (1025, 3072, 1024), (77, 8, 64)])
def test_gemm_bias(cuda, tile, M, N, K):
    from mj_video_amd import ops
    a, w, b = rnd(M, K, seed=1), rnd(N, K, std=0.05, seed=2), rnd(N, std=0.1, seed=3)
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w.to(cuda), out, ops.EPI_BIAS, bias=b.to(cuda))
    ref = (a.float() @ w.float().t() + b.float()).to(BF)
    # one rounding point, accumulation order differs: <= 1 ulp, and the vast majority bit-identical
    assert_close_bf16(out, ref, 1, frac_exact=0.98, atol=1e-6, what="gemm_bias")
    out2 = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w.to(cuda), out2, ops.EPI_BIAS)
    assert_close_bf16(out2, (a.float() @ w.float().t()).to(BF), 1, frac_exact=0.98, atol=1e-6, what="gemm_nobias")


@pytest.mark.parametrize("M,N,K", [(200, 136, 192), (513, 520, 64), (700, 256, 128), (256, 1024, 320), (2200, 768, 2048)])
def test_gemm_exact_integers(cuda, tile, M, N, K):
    """small-integer operands: every product and partial sum is exact in fp32, so the result must be bit-exact
    (catches any fragment-layout / swizzle / transposition / pipeline-race mistake; asymmetric A and W;
    K covers 1, 2, 3, 5 and 32 K-tiles of the software pipeline, M and N ragged against both tile sizes)"""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-4, 5, (M, K), generator=g).float().to(BF)
    w = torch.randint(-3, 4, (N, K), generator=g).float().to(BF)
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w.to(cuda), out, ops.EPI_BIAS)
    ref = (a.float() @ w.float().t()).to(BF)
    assert torch.equal(out.cpu(), ref)


def test_gemm_gelu_relu(cuda, tile):
    from mj_video_amd import ops
    M, N, K = 257, 512, 128
    a, w, b = rnd(M, K, seed=1), rnd(N, K, std=0.1, seed=2), rnd(N, std=0.1, seed=3)
    lin = (a.float() @ w.float().t() + b.float()).to(BF)
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w.to(cuda), out, ops.EPI_BIAS_GELU, bias=b.to(cuda))
    # two rounding points (Linear, GELU): a 1-ulp flip of the first can move the second by up to 2 ulps
    assert_close_bf16(out, F.gelu(lin.float()).to(BF), 2, frac_exact=0.97, atol=2e-3, what="gemm_gelu")
    ops.gemm(a.to(cuda), w.to(cuda), out, ops.EPI_BIAS_RELU, bias=b.to(cuda))
    assert_close_bf16(out, F.relu(lin), 1, frac_exact=0.98, atol=1e-6, what="gemm_relu")


def test_gemm_gelu_is_exact_for_every_bf16_input(cuda, tile):
    """the GELU epilogue is a table over the finite function bf16 -> bf16 (fast path: every lane of the wave inside the table;
    general path: x / 2 below it, x or -0 above): ALL 65 280 finite bf16 values go through it (compared on the 64 768 of them between 2^-125 and 2^127 in magnitude, and 0) as GEMM outputs (A holds the
    values, W is the identity, so every sum has one term and the Linear's rounding is exact) and must equal torch's bf16 GELU
    bit for bit - in natural order (whole waves inside / outside the table) and shuffled (mixed waves -> general path)."""
    from mj_video_amd import ops
    bits = torch.arange(65536, dtype=torch.int32)
    vals = bits.to(torch.int16).view(BF)
    finite = torch.isfinite(vals.float())
    vals = vals[finite]                       # 65 280 values
    pad = (-vals.numel()) % 64
    vals = torch.cat([vals, torch.zeros(pad, dtype=BF)])
    eye = torch.eye(64, dtype=BF)
    for order in ("natural", "shuffled"):
        v = vals if order == "natural" else vals[torch.randperm(vals.numel(), generator=torch.Generator().manual_seed(3))]
        a = v.view(-1, 64).contiguous()
        out = torch.empty(a.shape[0], 64, dtype=BF, device=cuda)
        ops.gemm(a.to(cuda), eye.to(cuda), out, ops.EPI_BIAS_GELU)
        ref = F.gelu(a)                        # torch's CPU bf16 GELU - what the reference's nn.GELU computes on bf16 activations
        got = out.cpu()
        same = got.view(torch.int16) == ref.view(torch.int16)
        # (+0 / -0 compare equal; inputs below 2^-125 - subnormal, or with a subnormal half - are left out: torch's CPU kernel
        # flushes subnormal results to 0, the table's x / 2 keeps them, and no activation gets there)
        same |= (got.float() == 0) & (ref.float() == 0)
        same |= a.float().abs() < 2.0 ** -125
        assert same.all(), (order, int((~same).sum()), a[~same][:8], got[~same][:8], ref[~same][:8])


def test_gemm_gelu_propagates_nonfinite_preactivations(cuda, tile):
    """NaN / +-inf pre-activations (the fast path's wave-wide range test once ran on fmaxf / fminf, which DROP a NaN
    operand, and gathered a finite value for it).  The values arrive through the BIAS (A = 0, so the sum is the bias exactly and a
    non-finite value stays in its own column): one NaN / inf column among in-table columns - the vote must fail on the lanes that
    hold it - then every non-finite bf16 pattern.  Must equal torch's CPU bf16 GELU: NaN where it gives NaN, bits elsewhere."""
    from mj_video_amd import ops
    M, N, K = 300, 64, 64
    a = torch.zeros(M, K, dtype=BF, device=cuda)
    w = rnd(N, K, seed=5).to(cuda)
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(BF)
    nonfinite = bits[~torch.isfinite(bits.float())]          # 2 infinities + 508 NaN patterns
    cases = []
    for special in (float("nan"), float("inf"), -float("inf"), 3.0e38, -3.0e38):
        for col in (0, 17, 63):
            b = rnd(N, std=0.7, seed=col)
            b[col] = special
            cases.append(b)
    pad = (-nonfinite.numel()) % N
    cases += list(torch.cat([nonfinite, torch.ones(pad, dtype=BF)]).view(-1, N))
    for b in cases:
        out = torch.empty(M, N, dtype=BF, device=cuda)
        ops.gemm(a, w, out, ops.EPI_BIAS_GELU, bias=b.to(cuda))
        ref = F.gelu(b).expand(M, N)
        got = out.cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), (b, got[0], ref[0])
        fin = ~torch.isnan(ref)
        assert torch.equal(got[fin].view(torch.int16), ref[fin].view(torch.int16)), (b, got[0], ref[0])


def test_gemm_scale_res_and_rowmaps(cuda, tile):
    from mj_video_amd import ops
    M, N, K = 320, 256, 128
    a, w, b = rnd(M, K, seed=1), rnd(N, K, std=0.1, seed=2), rnd(N, std=0.1, seed=3)
    ls, res = rnd(N, std=0.5, seed=4), rnd(M, N, seed=5)
    lin = (a.float() @ w.float().t() + b.float()).to(BF)
    ref = (res.float() + (lin.float() * ls.float()).to(BF).float()).to(BF)
    x = res.clone().to(cuda)
    ops.gemm(a.to(cuda), w.to(cuda), x, ops.EPI_SCALE_RES, bias=b.to(cuda), scale=ls.to(cuda), res=x)  # in place
    assert_close_bf16(x, ref, 2, frac_exact=0.97, atol=0.02, what="gemm_scale_res")
    # residual only (LLM wo / w2)
    x = res.clone().to(cuda)
    ops.gemm(a.to(cuda), w.to(cuda), x, ops.EPI_SCALE_RES, res=x)
    ref2 = (res.float() + (a.float() @ w.float().t()).to(BF).float()).to(BF)
    # a 1-ulp flip of the rounded GEMM term (|v| up to ~4 -> 2^-6) survives cancellation against the residual
    assert_close_bf16(x, ref2, 2, frac_exact=0.97, atol=0.02, what="gemm_res")
    # patch-embed style: residual row = 1 + m % 64 of a table, output skips one CLS slot per group of 64
    table = rnd(65, N, seed=6)
    out = torch.zeros(M + M // 64, N, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w.to(cuda), out, ops.EPI_SCALE_RES, bias=b.to(cuda), res=table.to(cuda), res_mod=64, res_off=1,
             out_group=64, out_pad=1)
    ref3 = (lin.float() + table[1:].repeat(M // 64, 1).float()).to(BF).view(M // 64, 64, N)
    got = out.cpu().view(M // 64, 65, N)
    assert_close_bf16(got[:, 1:], ref3, 2, frac_exact=0.97, atol=0.02, what="gemm_patch")
    assert (got[:, 0] == 0).all()
    # explicit output rows (splice)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(7)).to(torch.int32)
    out = torch.zeros(M, N, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w.to(cuda), out, ops.EPI_BIAS, bias=b.to(cuda), out_rows=perm.to(cuda))
    assert_close_bf16(out.cpu()[perm.long()], lin, 1, frac_exact=0.98, atol=1e-6, what="gemm_out_rows")


def test_gemm_silu_mul(cuda, tile):
    from mj_video_amd import ops
    M, FF, K = 530, 256, 128
    a, w1, w3 = rnd(M, K, seed=1), rnd(FF, K, std=0.1, seed=2), rnd(FF, K, std=0.1, seed=3)
    w13 = torch.stack([w1.view(FF // 16, 16, K), w3.view(FF // 16, 16, K)], dim=1).reshape(2 * FF, K).contiguous()
    out = torch.empty(M, FF, dtype=BF, device=cuda)
    ops.gemm(a.to(cuda), w13.to(cuda), out, ops.EPI_SILU_MUL)
    g = (a.float() @ w1.float().t()).to(BF)
    u = (a.float() @ w3.float().t()).to(BF)
    ref = F.silu(g) * u  # bf16 ops, as modeling_internlm2.py:262
    assert_close_bf16(out, ref, 3, frac_exact=0.95, atol=2e-3, what="gemm_silu_mul")


@pytest.mark.parametrize("M,N,K", [(1000, 512, 256), (65600 // 8, 1024, 1024), (300, 264, 64), (128, 256, 2048)])
def test_gemm_two_per_cu_kernel_equals_the_256_tile(cuda, M, N, K):
    """tile code 2 (round 6): the 128 x 256 x 32 kernel that runs two workgroups per CU - measured, not dispatched automatically
    (profiles/r06_d_gemm_two_per_cu_ab.txt: 23-36 % slower than the 256 x 256 kernel) - sums k in the same 32-deep MFMA order as the
    256-tile kernel, so its three epilogues (bias, bias + GELU on its 6-KiB sub-table, LayerScale + residual in place) must agree
    with it BIT FOR BIT, ragged last m-tile and a column count that is no multiple of 256 included; other epilogues are refused."""
    from mj_video_amd import ops, _lib
    a, w = rnd(M, K, seed=1).to(cuda), rnd(N, K, std=0.05, seed=2).to(cuda)
    bias, scale = rnd(N, std=0.2, seed=3).to(cuda), (1 + 0.1 * rnd(N, seed=4)).to(BF).to(cuda)
    for epi, kw in ((_lib.EPI_BIAS, dict(bias=bias)), (_lib.EPI_BIAS_GELU, dict(bias=bias)), (_lib.EPI_SCALE_RES, dict(bias=bias, scale=scale))):
        outs = []
        for tile in (256, 2):
            out = rnd(M, N, seed=5).to(cuda)        # (the residual stream for SCALE_RES, overwritten otherwise)
            ops.gemm(a, w, out, epi, res=out if epi == _lib.EPI_SCALE_RES else None, tile=tile, **kw)
            outs.append(out)
        assert torch.isfinite(outs[1].float()).all()
        assert torch.equal(outs[0], outs[1]), (epi, (outs[0].float() - outs[1].float()).abs().max().item())
    x = (rnd(M, K, std=3.0, seed=7)).to(cuda)       # GELU over a wide range of pre-activations (|x| up to ~50 and tiny ones)
    x[:, ::7] *= 1e-4
    o256, o2 = torch.empty(M, N, dtype=BF, device=cuda), torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(x, w, o256, _lib.EPI_BIAS_GELU, bias=bias, tile=256)
    ops.gemm(x, w, o2, _lib.EPI_BIAS_GELU, bias=bias, tile=2)
    assert torch.equal(o256, o2)
    with pytest.raises(_lib.MjvLibraryError, match="tile 2"):
        ops.gemm(a, w, torch.empty(M, N, dtype=BF, device=cuda), _lib.EPI_BIAS_RELU, bias=bias, tile=2)


@pytest.mark.parametrize("K", [256, 1024])
@pytest.mark.parametrize("epi", ["bias", "silu_mul"])
def test_gemm_persistent_equals_one_tile_kernel(cuda, epi, K):
    """round 3: launches of the plain-bias / SiLU epilogues with more 256^2 tiles than CUs run the PERSISTENT kernel
    (gemm256p_kernel: tiles walked by one workgroup per CU, the next tile's first K-tile prefetched under a two-half epilogue).
    Same MFMA order and rounding points as the one-tile kernel, so: 16 384 x 2 048 outputs in ONE launch (512 tiles ->
    persistent) must equal, bit for bit, the same rows computed as two launches of 256 tiles each (one-tile kernel) - and the
    integer-data case must be exact."""
    from mj_video_amd import ops
    _persistent_case(cuda, epi, K, 16384, 2048, [(0, 8192), (8192, 16384)])
    # ragged: the last m-tile has 100 rows and the last n-tile 24 columns (bias) - clamped loads, masked stores in the walk too
    if epi == "bias":
        _persistent_case(cuda, epi, K, 16484, 2072, [(0, 7168), (7168, 14336), (14336, 16484)], exact=False)   # (<= 252 tiles per chunk)


def _persistent_case(cuda, epi, K, M, N, chunks, exact=True):
    from mj_video_amd import ops
    a = rnd(M, K, seed=11).to(cuda)
    w = rnd(N, K, std=0.05, seed=12).to(cuda)
    b = rnd(N, std=0.1, seed=13).to(cuda)
    code = ops.EPI_BIAS if epi == "bias" else ops.EPI_SILU_MUL
    cols = N if epi == "bias" else N // 2
    kw = dict(bias=b) if epi == "bias" else {}
    one = torch.empty(M, cols, dtype=BF, device=cuda)
    ops.gemm(a, w, one, code, tile=256, **kw)
    two = torch.empty(M, cols, dtype=BF, device=cuda)
    for lo, hi in chunks:
        ops.gemm(a[lo:hi], w, two[lo:hi], code, tile=256, **kw)
    assert torch.equal(one, two)
    if epi == "bias" and exact:
        g = torch.Generator().manual_seed(5)
        ai = torch.randint(-4, 5, (M, K), generator=g).float().to(BF)
        wi = torch.randint(-3, 4, (N, K), generator=g).float().to(BF)
        ops.gemm(ai.to(cuda), wi.to(cuda), one, ops.EPI_BIAS, tile=256)
        assert torch.equal(one.cpu(), (ai.float() @ wi.float().t()).to(BF))


# ---------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("rows,dim", [(5, 128), (1025, 1024), (300, 4096)])
def test_layernorm(cuda, rows, dim):
    from mj_video_amd import ops
    x, g, b = rnd(rows, dim, seed=1), (1 + 0.1 * torch.randn(dim)).to(BF), (0.1 * torch.randn(dim)).to(BF)
    out = torch.empty(rows, dim, dtype=BF, device=cuda)
    ops.layernorm(x.to(cuda), g.to(cuda), b.to(cuda), out, 1e-6)
    ref = F.layer_norm(x.float(), (dim,), g.float(), b.float(), 1e-6).to(BF)
    assert_close_bf16(out, ref, 1, frac_exact=0.98, atol=1e-6, what="layernorm")


def test_layernorm_pixel_shuffle(cuda):
    from mj_video_amd import ops
    tiles, G, d = 3, 8, 128
    x = rnd(tiles * (G * G + 1), d, seed=1)
    g, b = (1 + 0.1 * torch.randn(4 * d)).to(BF), (0.1 * torch.randn(4 * d)).to(BF)
    out = torch.empty(tiles * (G // 2) ** 2, 4 * d, dtype=BF, device=cuda)
    ops.layernorm(x.to(cuda), g.to(cuda), b.to(cuda), out, 1e-5, gather_grid=G)
    # independent restatement of modeling_internvl_chat.py:228-242,255-260 with tensor ops
    v = x.view(tiles, G * G + 1, d)[:, 1:, :].reshape(tiles, G, G, d)
    n, w_, h_, c = v.shape
    v = v.view(n, w_, h_ // 2, c * 2).permute(0, 2, 1, 3).contiguous()
    v = v.view(n, h_ // 2, w_ // 2, c * 4).permute(0, 2, 1, 3).contiguous().view(n, -1, 4 * d)
    ref = F.layer_norm(v.float(), (4 * d,), g.float(), b.float(), 1e-5).to(BF).view(-1, 4 * d)
    assert_close_bf16(out, ref, 1, frac_exact=0.98, atol=1e-6, what="layernorm_pixshuf")


def test_rmsnorm_and_gather(cuda):
    from mj_video_amd import ops
    rows, dim = 333, 2048
    x, w = rnd(rows, dim, seed=1), (1 + 0.1 * torch.randn(dim)).to(BF)

    def ref_fn(t):
        h = t.float()
        h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + 1e-5)
        return w * h.to(BF)

    out = torch.empty(rows, dim, dtype=BF, device=cuda)
    ops.rmsnorm(x.to(cuda), w.to(cuda), out, 1e-5)
    assert_close_bf16(out, ref_fn(x), 1, frac_exact=0.98, atol=1e-6, what="rmsnorm")
    idx = torch.tensor([5, 0, 332, 17], dtype=torch.int32)
    out = torch.empty(4, dim, dtype=BF, device=cuda)
    ops.rmsnorm(x.to(cuda), w.to(cuda), out, 1e-5, row_index=idx.to(cuda))
    assert_close_bf16(out, ref_fn(x[idx.long()]), 1, frac_exact=0.98, atol=1e-6, what="rmsnorm_gather")


def test_rope_split(cuda):
    from mj_video_amd import ops
    rows, KV, G, D = 70, 2, 2, 128
    qkv = rnd(rows, KV * (G + 2) * D, seed=1)
    pos = torch.arange(rows, dtype=torch.int32) % 50
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(64).float(), inv)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos().to(BF), emb.sin().to(BF)
    q = torch.empty(rows, KV * G * D, dtype=BF, device=cuda)
    k = torch.empty(rows, KV * D, dtype=BF, device=cuda)
    ops.rope_split(qkv.to(cuda), q, k, cos.to(cuda), sin.to(cuda), pos.to(cuda), KV, G)
    t = qkv.view(rows, KV, G + 2, D)

    def rot(x):
        return torch.cat((-x[..., D // 2:], x[..., :D // 2]), -1)

    c, s = cos[pos.long()][:, None, None, :], sin[pos.long()][:, None, None, :]
    qr = (t[:, :, :G] * c) + (rot(t[:, :, :G]) * s)          # bf16 ops: modeling_internlm2.py:240-247
    kr = (t[:, :, G:G + 1] * c) + (rot(t[:, :, G:G + 1]) * s)
    assert torch.equal(q.cpu(), qr.reshape(rows, -1))           # elementwise bf16 arithmetic: bit-exact
    assert torch.equal(k.cpu(), kr.reshape(rows, -1))


@pytest.mark.parametrize("M", [17488, 2186, 300])
def test_gemm_rope_qkv_epilogue_equals_gemm_then_rope(cuda, M):
    """InternLM2 wqkv with RoPE + GQA de-interleave fused into the GEMM epilogue (MJV_EPI_ROPE_QKV) against the two-kernel
    path it replaces (plain Linear, then rope_split - itself bit-exact against the torch formula, test_rope_split): q, k
    and the v columns must be bit-identical.  M = 17488 takes the 256^2 kernel + a peeled tail (which falls back to the
    two-kernel path inside the library), 2186 the 256^2 kernel alone, 300 the small kernels only."""
    from mj_video_amd import ops
    H, KV, D, hid = 16, 8, 128, 2048
    G = H // KV
    N = (H + 2 * KV) * D
    a, w = rnd(M, hid, seed=1).to(cuda), rnd(N, hid, std=0.03, seed=2).to(cuda)
    g = torch.Generator().manual_seed(4)
    pos = torch.randint(0, 4096, (M,), generator=g).to(torch.int32).to(cuda)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(4096).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos().to(BF).to(cuda).contiguous(), emb.sin().to(BF).to(cuda).contiguous()
    qkv0 = torch.zeros(M, N, dtype=BF, device=cuda)
    q0, k0 = torch.empty(M, H * D, dtype=BF, device=cuda), torch.empty(M, KV * D, dtype=BF, device=cuda)
    ops.gemm(a, w, qkv0, ops.EPI_BIAS)
    ops.rope_split(qkv0, q0, k0, cos, sin, pos, KV, G)
    qkv1 = torch.zeros(M, N, dtype=BF, device=cuda)
    q1, k1 = torch.empty(M, H * D, dtype=BF, device=cuda), torch.empty(M, KV * D, dtype=BF, device=cuda)
    ops.gemm(a, w, qkv1, ops.EPI_ROPE_QKV, rope=(cos, sin, pos, q1, k1, G))
    torch.cuda.synchronize()
    assert torch.equal(q0, q1) and torch.equal(k0, k1)
    v0 = qkv0.view(M, KV, G + 2, D)[:, :, G + 1]
    v1 = qkv1.view(M, KV, G + 2, D)[:, :, G + 1]
    assert torch.equal(v0, v1)


def test_patchify_cls_embed(cuda):
    from mj_video_amd import ops
    tiles, S, P = 2, 56, 14
    px = rnd(tiles, 3, S, S, seed=1)
    kp = 640
    out = torch.empty(tiles * 16, kp, dtype=BF, device=cuda)
    ops.patchify(px.to(cuda), out, P)
    ref = F.unfold(px.float(), kernel_size=P, stride=P).transpose(1, 2).reshape(tiles * 16, 3 * P * P).to(BF)
    assert torch.equal(out.cpu()[:, :588], ref)
    assert (out.cpu()[:, 588:] == 0).all()
    x = torch.zeros(tiles * 17, 128, dtype=BF, device=cuda)
    cls, p0 = rnd(128, seed=2), rnd(128, seed=3)
    ops.cls_rows(x, cls.to(cuda), p0.to(cuda), tiles, 17)
    assert torch.equal(x.cpu()[0], cls + p0) and torch.equal(x.cpu()[17], cls + p0) and (x.cpu()[1:17] == 0).all()
    table = rnd(1000, 256, seed=4)
    ids = torch.tensor([3, 999, 7, 7, 0, 500], dtype=torch.int32)
    xo = torch.zeros(6, 256, dtype=BF, device=cuda)
    ops.embed_gather(ids.to(cuda), table.to(cuda), xo, 7)
    exp = table[ids.long()].clone()
    exp[2:4] = 0
    assert torch.equal(xo.cpu(), exp)


# ------------------------------------------------------------------------------------------ attention
@pytest.fixture(params=[0, 4, 5, 6, 7])
def attn_variant(request, cuda):
    """attention kernel selector: 0 = automatic (round 3: the two-sub-block pipelined kernel where instantiated), 4 = the
    register-staged round-1 kernel for every shape, 5 = the round-2 choice (LDS-DMA staging up to 4096 keys), 6 / 7 = the
    round-3 kernel with two / four waves per workgroup (automatic = four; the two must agree bit for bit)"""
    from mj_video_amd import ops
    ops.attention_set_variant(request.param)
    yield request.param
    ops.attention_set_variant(0)


def two_wave_form_refused(cuda, variant, D):
    """kernel 6 (two waves per workgroup) exists at head_dim 64 only (include/mjv.h): at 128 it would run one wave per SIMD and
    is refused with MJV_E_UNSUPPORTED.  Returns True after checking the refusal, so the calling test ends there."""
    if variant != 6 or D != 128:
        return False
    from mj_video_amd import ops, _lib
    q = torch.zeros(64, D, dtype=BF, device=cuda)
    cu = torch.tensor([0, 64], dtype=torch.int32, device=cuda)
    with pytest.raises(_lib.MjvLibraryError, match="kernel 6"):
        ops.attention(q, q, q, torch.empty_like(q), cu, 64, 1, 1, D, True, D ** -0.5, 1, kernel=6)
    return True


def attn_reference(q, k, v, lens, H, G, D, causal, scale, mode):
    """fp32 reference with the reference's score rounding; q [N, H*D], k/v [N, (H/G)*D] packed."""
    out = torch.zeros(q.shape[0], H * D)
    s0 = 0
    for L in lens:
        for h in range(H):
            qh = q[s0:s0 + L, h * D:(h + 1) * D].float()
            kh = k[s0:s0 + L, (h // G) * D:(h // G + 1) * D].float()
            vh = v[s0:s0 + L, (h // G) * D:(h // G + 1) * D].float()
            sc = qh @ kh.t()
            if mode == 2:      # the reference's flash-attention numerics: fp32 scores, never rounded
                sc = sc * scale
            else:
                sc = (sc.to(BF).float() * scale).to(BF).float() if mode else (sc * scale).to(BF).float()
            if causal:
                sc = sc.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
            p = torch.softmax(sc, -1).to(BF).float()
            out[s0:s0 + L, h * D:(h + 1) * D] = p @ vh
        s0 += L
    return out.to(BF)


@pytest.mark.parametrize("D,H,G,causal,lens", [
    (64, 2, 1, False, [17, 17, 17]),
    (64, 16, 1, False, [1025, 1025]),
    (64, 2, 1, False, [257, 64, 129]),
    (128, 2, 2, True, [150]),
    (128, 4, 2, True, [650, 131, 64, 1]),
    (128, 16, 2, True, [2186]),
])
def test_attention(cuda, attn_variant, D, H, G, causal, lens):
    from mj_video_amd import ops
    if two_wave_form_refused(cuda, attn_variant, D):
        return
    N = sum(lens)
    KVH = H // G
    q, k, v = rnd(N, H * D, seed=1), rnd(N, KVH * D, seed=2), rnd(N, KVH * D, seed=3)
    scale = D ** -0.5
    mode = 1 if causal else 0
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    out = torch.empty(N, H * D, dtype=BF, device=cuda)
    ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu.to(cuda), max(lens), H, G, D, causal, scale, mode)
    ref = attn_reference(q, k, v, lens, H, G, D, causal, scale, mode)
    # P is rounded to bf16 before (ours) vs after (reference) normalisation: independent relative errors of
    # std 2^-9/sqrt(3) on every term in both, plus the two final roundings -> expected relative L2 distance
    # ~2.3e-3; bound 4e-3 (= 2^-8).  Element-wise: 2 ulps, or 0.02 absolute where terms cancel.
    err = (out.float().cpu() - ref.float()).abs()
    assert torch.isfinite(out.float()).all()
    rel = (out.float().cpu() - ref.float()).norm() / ref.float().norm()
    assert rel.item() < 4e-3, f"relative L2 error {rel.item():.3e}"
    assert err.max().item() < 0.03, f"max abs err {err.max().item()}"
    assert_close_bf16(out, ref, 2, atol=0.02, what="attention")


@pytest.mark.parametrize("kernel", [0, 6])
@pytest.mark.parametrize("D,H,G,causal,lens", [
    (64, 2, 1, False, [17, 17, 17]),
    (64, 16, 1, False, [1025, 1025]),
    (64, 2, 1, False, [257, 64, 129]),
    (64, 2, 1, True, [300, 65]),
    (128, 2, 2, True, [150]),
    (128, 4, 2, True, [650, 131, 64, 1]),
    (128, 16, 2, True, [2186]),
    (128, 4, 2, False, [577]),
])
def test_attention_unrounded_scores_mode2(cuda, kernel, D, H, G, causal, lens):
    """score_round_mode 2 = the reference's flash-attention numerics (modeling_intern_vit.py:229-244, modeling_internlm2.py:
    437-561: fp32 softmax on unrounded scores), round-3 kernel only: against the fp32 reference WITHOUT the score rounding, same
    bounds as test_attention; and strictly closer to that reference than the eager-numerics result is (the rounding is really
    gone); the older kernels refuse the mode."""
    from mj_video_amd import ops, _lib
    if two_wave_form_refused(cuda, kernel, D):
        return
    N = sum(lens)
    KVH = H // G
    q, k, v = rnd(N, H * D, seed=1), rnd(N, KVH * D, seed=2), rnd(N, KVH * D, seed=3)
    scale = D ** -0.5
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    out = torch.empty(N, H * D, dtype=BF, device=cuda)
    ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu.to(cuda), max(lens), H, G, D, causal, scale, 2, kernel=kernel)
    ref = attn_reference(q, k, v, lens, H, G, D, causal, scale, 2)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    rel = ((o - ref.float()).norm() / ref.float().norm()).item()
    assert rel < 4e-3, f"relative L2 error {rel:.3e}"
    assert_close_bf16(out, ref, 2, atol=0.02, what="attention mode 2")
    eager = torch.empty_like(out)
    ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), eager, cu.to(cuda), max(lens), H, G, D, causal, scale, 1 if causal else 0, kernel=kernel)
    rel_eager = ((eager.float().cpu() - ref.float()).norm() / ref.float().norm()).item()
    if max(lens) >= 64:
        assert rel <= rel_eager, (rel, rel_eager)
    with pytest.raises(_lib.MjvLibraryError, match="score_round_mode 2"):
        ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu.to(cuda), max(lens), H, G, D, causal, scale, 2, kernel=5)


@pytest.mark.parametrize("D,causal,L", [(64, False, 1025), (64, False, 1024), (64, True, 700), (128, True, 2186), (128, False, 577)])
@pytest.mark.parametrize("ramp", [0.6, 0.12, -0.6])
def test_attention_rising_scores_mode2(cuda, D, causal, L, ramp):
    """the offset-raise ramps of the test below through the unrounded-score mode (the raise logic is shared, the rounding of the
    maximum that sets the new offset is not)"""
    _rising_scores_case(cuda, D, causal, L, ramp, 2)


@pytest.mark.parametrize("D,causal,L", [(64, False, 1025), (64, False, 1024), (64, True, 700), (128, True, 2186), (128, False, 577)])
@pytest.mark.parametrize("ramp", [0.6, 0.12, -0.6])
def test_attention_rising_scores_exercise_the_offset_raise(cuda, attn_variant, D, causal, L, ramp):
    """The round-3 kernel keeps an integer softmax offset per query and raises it only when a lane's partial row sum reaches
    2^20 - on bounded random data that branch fires for the first unit only (offset from 'none' to the first maximum), so
    random tests say nothing about a raise in mid-sequence, where O, l and the exponentials in flight all have to move to the
    new offset exactly once.  Here the scores RISE along the keys (ramp 0.6 / 0.12 log2 units per key plus noise: a raise every
    ~30 / ~150 keys, each by a few binades, so what was accumulated before stays significant after it), or fall (-0.6: never a
    raise, the first keys dominate).  Against the fp32 reference, same bounds as test_attention."""
    if two_wave_form_refused(cuda, attn_variant, D):
        return
    _rising_scores_case(cuda, D, causal, L, ramp, 1 if causal else 0)


def _rising_scores_case(cuda, D, causal, L, ramp, mode):
    from mj_video_amd import ops
    if not causal and L > 600 and ramp > 0.5:
        # (every query sees the top of the ramp: raw scores of +-3 400, where one bf16 ulp of the reference's OWN score rounding is
        # 2 in the exponent - all five kernel choices land at the same 7e-3 from the reference there; the ramp is halved instead)
        ramp = 0.3
    H, G = 4, 2 if D == 128 else 1
    KVH = H // G
    g = torch.Generator().manual_seed(41)
    scale = D ** -0.5
    # q = a fixed direction u (+ noise), k_j = (ramp * j / (scale * log2 e)) * u / |u|^2 (+ noise): q . k_j * scale * log2 e ~ ramp * j
    u = torch.randn(D, generator=g)
    u = u / u.norm()
    qn = 4.0
    q = (qn * u + 0.3 * torch.randn(L, H, D, generator=g)).to(BF)
    step = ramp / (scale * math.log2(math.e) * qn)
    j = torch.arange(L, dtype=torch.float32).view(L, 1, 1)
    k = (step * j * u + 0.3 * torch.randn(L, KVH, D, generator=g)).to(BF)
    v = torch.randn(L, KVH, D, generator=g).to(BF)
    q2, k2, v2 = q.reshape(L, H * D), k.reshape(L, KVH * D), v.reshape(L, KVH * D)
    cu = torch.tensor([0, L], dtype=torch.int32)
    out = torch.full((L, H * D), float("nan"), dtype=BF, device=cuda)
    ops.attention(q2.to(cuda), k2.to(cuda), v2.to(cuda), out, cu.to(cuda), L, H, G, D, causal, scale, mode)
    ref = attn_reference(q2, k2, v2, [L], H, G, D, causal, scale, mode)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    rel = (o - ref.float()).norm() / ref.float().norm()
    # (large raw scores: a one-ulp flip of the reference's own bf16 score rounding moves single probabilities by several per
    # cent - tools/fuzz_kernels.py - so the cell bound is relative to the largest output)
    assert rel.item() < 6e-3, f"relative L2 error {rel.item():.3e}"
    assert (o - ref.float()).abs().max().item() < 0.08 * ref.float().abs().max().item() + 0.02


def test_attention_exact_selection(cuda, attn_variant):
    """one-hot softmax (a huge score on one key): output must equal that key's V row bit for bit; checks the
    key<->value pairing of the transposed LDS reads and the causal/ragged masks"""
    from mj_video_amd import ops
    D, H, L = 128, 2, 200
    if two_wave_form_refused(cuda, attn_variant, D):
        return
    g = torch.Generator().manual_seed(3)
    v = torch.randn(L, D, generator=g).to(BF)
    target = torch.randint(0, L, (L,), generator=g)
    target = torch.minimum(target, torch.arange(L))          # causal: only keys <= query
    # key i = e_(i mod 64) + e_(64 + i div 64): unique per key, so q = 2048 * key_target scores 4096 on the
    # target, <= 2048 on every other key -> softmax is one-hot after the 1/sqrt(D) scaling
    kk = torch.zeros(L, D)
    for i in range(L):
        kk[i, i % 64] = 1.0
        kk[i, 64 + i // 64] = 1.0
    qq = torch.zeros(L, D)
    for i in range(L):
        t = int(target[i])
        qq[i, t % 64] = 2048.0
        qq[i, 64 + t // 64] = 2048.0
    q = qq.to(BF).repeat(1, H)
    k = kk.to(BF)
    cu = torch.tensor([0, L], dtype=torch.int32)
    out = torch.empty(L, H * D, dtype=BF, device=cuda)
    ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu.to(cuda), L, H, H, D, True, 1.0 / math.sqrt(D), 1)
    exp = v[target]
    assert torch.equal(out.cpu()[:, :D], exp) and torch.equal(out.cpu()[:, D:], exp)


@pytest.mark.parametrize("M,N,K,epi", [(1104, 2048, 8192, "scale_res"), (64, 1024, 4096, "scale_res"), (64, 3072, 4096, "bias"),
                                       (80, 1024, 4096, "silu"), (8, 1024, 8192, "relu"), (300, 136, 4160, "bias")])
def test_gemm_split_k_matches_unsplit(cuda, M, N, K, epi):
    """under-filled 128-tile launches (peeled tail rows, batch-sized head GEMMs) split K through the caller's workspace when
    one is given: operands in {-1,0,1} keep every partial sum an exactly representable integer, so the split result must
    equal the unsplit one and the fp32 reference bit for bit, for every epilogue the finishing kernel runs"""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(13)
    a = torch.randint(-1, 2, (M, K), generator=g).float().to(BF).to(cuda)
    w = torch.randint(-1, 2, (N, K), generator=g).float().to(BF).to(cuda)
    # thin out the operands so |sum| stays below 256 (exact in bf16) even at K = 8192
    a = a * (torch.rand(M, K, generator=g) < 0.15).to(BF).to(cuda)
    bias = torch.randint(-2, 3, (N,), generator=g).float().to(BF).to(cuda)
    res = torch.randint(-8, 9, (M, N if epi != "silu" else N // 2), generator=g).float().to(BF).to(cuda)
    kw = {"bias": dict(epilogue=ops.EPI_BIAS, bias=bias), "relu": dict(epilogue=ops.EPI_BIAS_RELU, bias=bias),
          "scale_res": dict(epilogue=ops.EPI_SCALE_RES, bias=bias, res=res), "silu": dict(epilogue=ops.EPI_SILU_MUL)}[epi]
    nout = N // 2 if epi == "silu" else N
    outs = []
    ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=cuda)
    try:
        ops.gemm_set_tile(128)    # the shapes above reach the 128-tile kernel as peeled tails inside the model
        for use_ws in (False, True):
            ops.set_gemm_workspace(ws if use_ws else None)
            ws.fill_(255)         # all-ones words are NaNs: a slice that was written is finite afterwards
            out = torch.full((M, nout), 7.0, dtype=BF, device=cuda)
            ops.gemm(a, w, out, **kw)
            outs.append(out.clone())
            touched = bool(torch.isfinite(ws[:65536].view(torch.float32)).all())
            assert touched == use_ws, "split-K path taken when it should not be (or not taken when it should)"
    finally:
        ops.gemm_set_tile(0)
        ops.set_gemm_workspace(None)
    # the scratch can also travel in the call itself (thread-safe form): same result
    out = torch.full((M, nout), 7.0, dtype=BF, device=cuda)
    ops.gemm(a, w, out, workspace=ws, tile=128, **kw)
    outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    lin = a.float() @ w.float().t()
    if epi == "bias":
        ref = (lin + bias.float()).to(BF)
    elif epi == "relu":
        ref = F.relu((lin + bias.float()).to(BF))
    elif epi == "scale_res":
        ref = (res.float() + (lin + bias.float()).to(BF).float()).to(BF)
    else:
        lin16 = lin.to(BF).view(M, N // 32, 2, 16)
        ref = (F.silu(lin16[:, :, 0]) * lin16[:, :, 1]).reshape(M, N // 2)
        assert_close_bf16(outs[1], ref, 3, frac_exact=0.95, atol=2e-3, what="split_k_silu")
        return
    assert torch.equal(outs[1], ref)


def test_gemm_pipeline_race_screen(cuda):
    """the 256^2 kernel keeps LDS-DMA in flight across barriers: repeat an exact-integer problem many times on a
    busy chip (many tiles, deep K) and require every run bit-identical to the exact result"""
    from mj_video_amd import ops
    ops.gemm_set_tile(256)
    try:
        # 136 tiles: the one-tile-per-workgroup kernel; 396 tiles (> 256 CUs, 16 K-tiles): the PERSISTENT kernel, whose first
        # wait per tile is a counted vmcnt that leaves the previous tile's stores in flight (its only guard was a
        # bit-equality test at two shapes)
        for M, N, K in ((4096 + 37, 2048, 1024), (8192 + 37, 3072, 1024), (8192, 3072, 256)):
            g = torch.Generator().manual_seed(11)
            a = torch.randint(-3, 4, (M, K), generator=g).float().to(BF).to(cuda)
            w = torch.randint(-3, 4, (N, K), generator=g).float().to(BF).to(cuda)
            ref = (a.float() @ w.float().t()).to(BF)
            out = torch.empty(M, N, dtype=BF, device=cuda)
            for it in range(20):
                out.zero_()
                ops.gemm(a, w, out, ops.EPI_BIAS)
                assert torch.equal(out, ref), f"{M}x{N}x{K} iteration {it}: {(out != ref).sum().item()} wrong elements"
        # the persistent SiLU-mul form (8 stores per tile behind the prefetch instead of 16)
        M, ff, K = 8192 + 64, 2048, 512
        g = torch.Generator().manual_seed(12)
        a = torch.randint(-2, 3, (M, K), generator=g).float().to(BF).to(cuda)
        w1 = torch.randint(-2, 3, (ff, K), generator=g).float().to(BF)
        w3 = torch.randint(-2, 3, (ff, K), generator=g).float().to(BF)
        w13 = torch.stack([w1.view(ff // 16, 16, K), w3.view(ff // 16, 16, K)], dim=1).reshape(2 * ff, K).contiguous().to(cuda)
        gg, uu = (a.float().cpu() @ w1.float().t()).to(BF), (a.float().cpu() @ w3.float().t()).to(BF)
        ref = (F.silu(gg) * uu)
        first = torch.empty(M, ff, dtype=BF, device=cuda)
        ops.gemm(a, w13, first, ops.EPI_SILU_MUL)
        assert_close_bf16(first, ref, 3, frac_exact=0.95, atol=2e-3, what="persistent silu")
        out = torch.empty_like(first)
        for it in range(20):
            out.zero_()
            ops.gemm(a, w13, out, ops.EPI_SILU_MUL)
            assert torch.equal(out, first), f"persistent silu iteration {it}"
    finally:
        ops.gemm_set_tile(0)


@pytest.mark.parametrize("M,N,K,epi", [(1104, 2048, 8192, "scale_res"), (1104, 4096, 4096, "bias"), (2186, 2048, 8192, "scale_res"),
                                       (600, 1024, 4096, "silu"), (400, 520, 4160, "relu")])
def test_gemm_split_k_on_256_tiles(cuda, M, N, K, epi):
    """under-filled problems with deep K (the ~1100-row tails of the language tower, the whole GEMMs of a single-video
    forward) run as K slices of 256 x 256 tiles when the caller gives a workspace; exact-integer operands: the sliced result
    equals the unsliced one and the fp32 reference bit for bit, for every epilogue of the finishing kernel"""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(19)
    a = torch.randint(-1, 2, (M, K), generator=g).float().to(BF).to(cuda)
    w = torch.randint(-1, 2, (N, K), generator=g).float().to(BF).to(cuda)
    a = a * (torch.rand(M, K, generator=g) < 0.15).to(BF).to(cuda)
    bias = torch.randint(-2, 3, (N,), generator=g).float().to(BF).to(cuda)
    res = torch.randint(-8, 9, (M, N if epi != "silu" else N // 2), generator=g).float().to(BF).to(cuda)
    kw = {"bias": dict(epilogue=ops.EPI_BIAS, bias=bias), "relu": dict(epilogue=ops.EPI_BIAS_RELU, bias=bias),
          "scale_res": dict(epilogue=ops.EPI_SCALE_RES, bias=bias, res=res), "silu": dict(epilogue=ops.EPI_SILU_MUL)}[epi]
    nout = N // 2 if epi == "silu" else N
    ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=cuda)
    outs = []
    for tile in (0, 128):                  # automatic choice (K-sliced 256 tiles) / the 128-tile kernels with their own split-K
        ws.fill_(255)
        out = torch.full((M, nout), 7.0, dtype=BF, device=cuda)
        ops.gemm(a, w, out, workspace=ws, tile=tile, **kw)
        outs.append(out.clone())
        if tile == 0:                      # a whole 256 x 256 fp32 slice image is finite only if the sliced path wrote it
            assert bool(torch.isfinite(ws[:262144].view(torch.float32)).all()), "256-tile split-K path not taken"
    assert torch.equal(outs[0], outs[1])
    lin = a.float() @ w.float().t()
    if epi == "bias":
        ref = (lin + bias.float()).to(BF)
    elif epi == "relu":
        ref = F.relu((lin + bias.float()).to(BF))
    elif epi == "scale_res":
        ref = (res.float() + (lin + bias.float()).to(BF).float()).to(BF)
    else:
        lin16 = lin.to(BF).view(M, N // 32, 2, 16)
        ref = (F.silu(lin16[:, :, 0]) * lin16[:, :, 1]).reshape(M, N // 2)
        assert_close_bf16(outs[0], ref, 3, frac_exact=0.95, atol=2e-3, what="split_k256_silu")
        return
    assert torch.equal(outs[0], ref)


def test_attention_dma_staging_race_screen(cuda):
    """K / V tiles go to LDS by LDS-DMA into two buffers with one barrier per key tile.  Two screens on the model's shapes, a
    ragged shape and GQA causal shapes, each on a chip kept busy by a GEMM on another stream (a missing wait or barrier shows
    up as a changing result): (a) the round-2 kernel with LDS-DMA staging (variant 5) against its register-staged form
    (variant 4), which computes the same arithmetic in the same order: bit for bit; (b) the automatic choice (round 3: the
    two-sub-block pipelined kernel where it is instantiated, whose DMA is inline assembly behind hand-placed waits): repeated
    launches must reproduce the first one bit for bit, and stay within the parity tolerance of the round-2 kernel."""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(23)
    side = torch.cuda.Stream()
    big_a = torch.randn(8192, 2048, device=cuda).to(BF)
    big_w = torch.randn(4096, 2048, device=cuda).to(BF)
    big_o = torch.empty(8192, 4096, dtype=BF, device=cuda)
    for (n_seq, L, H, G, D, causal, mode) in [(16, 1025, 16, 1, 64, False, 0), (4, 2186, 16, 2, 128, True, 1), (3, 130, 16, 1, 64, False, 0),
                                              (2, 4096, 8, 2, 128, True, 1), (5, 33, 4, 1, 64, True, 0), (6, 257, 16, 1, 64, False, 0)]:
        N = n_seq * L
        q = torch.randn(N, H * D, generator=g).to(BF).to(cuda)
        k = torch.randn(N, (H // G) * D, generator=g).to(BF).to(cuda)
        v = torch.randn(N, (H // G) * D, generator=g).to(BF).to(cuda)
        cu = torch.arange(0, (n_seq + 1) * L, L, dtype=torch.int32, device=cuda)
        outs = {}
        try:
            for var in (4, 5, 5, 5, 0, 0, 0, 6, 7, 6, 7) if D == 64 else (4, 5, 5, 5, 0, 0, 0, 7, 7):   # (6: head_dim 64 only)
                ops.attention_set_variant(var)
                with torch.cuda.stream(side):
                    ops.gemm(big_a, big_w, big_o, ops.EPI_BIAS)
                o = torch.zeros(N, H * D, dtype=BF, device=cuda)
                ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode)
                outs.setdefault(var, []).append(o)
        finally:
            ops.attention_set_variant(0)
        torch.cuda.synchronize()
        for i, o in enumerate(outs[5]):
            assert torch.equal(o, outs[4][0]), f"D={D} L={L} causal={causal}: DMA launch {i} differs from the register-staged kernel"
        # (a query's arithmetic does not depend on how many waves share its workgroup: two- and four-wave blocks agree bit for bit)
        for i, o in enumerate(outs[0][1:] + outs.get(6, []) + outs[7]):
            assert torch.equal(o, outs[0][0]), f"D={D} L={L} causal={causal}: launch {i + 1} of the round-3 kernel differs from launch 0"
        rel = (outs[0][0].float() - outs[4][0].float()).norm() / outs[4][0].float().norm()
        assert rel.item() < 4e-3, f"D={D} L={L}: automatic kernel vs round-2 kernel relative L2 {rel.item():.2e}"


def test_attention_repeat_launch_stability(cuda):
    """round 3: every wave of the query-0 block of a peeled sequence reaches the output stage, and only one of them holds the
    query (a second writer of the same row lost the write-write race about once in 20 000 blocks before it was masked): 60
    launches on the vision tower's shape with q / k / v as column slices of one qkv buffer and the output pre-filled with NaN,
    every one finite and bit-identical to the first."""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(29)
    n_seq, L, H, D = 32, 1025, 16, 64
    N = n_seq * L
    qkv = torch.randn(N, 3 * H * D, generator=g).to(BF).to(cuda)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    cu = torch.arange(0, (n_seq + 1) * L, L, dtype=torch.int32, device=cuda)
    first = None
    for it in range(60):
        o = torch.full((N, H * D), float("nan"), dtype=BF, device=cuda)
        ops.attention(q, k, v, o, cu, L, H, 1, D, False, D ** -0.5, 0)
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all(), f"launch {it}: non-finite / unwritten output cells"
        if first is None:
            first = o
        else:
            assert torch.equal(o, first), f"launch {it} differs from launch 0"


def test_gemm_skinny_pipeline_race_screen(cuda):
    """the 64 x 32 kernel keeps two K-tiles of LDS-DMA in flight across its one barrier per K-tile (counted vmcnt, three
    buffers): repeat exact-integer problems of the shapes it serves (tails of 64 / 80 rows, deep and shallow K, the gating
    layers' handful of rows) on a chip kept busy by a large GEMM on another stream; every run bit-identical to the exact result"""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(17)
    side = torch.cuda.Stream()
    big_a = torch.randn(8192, 2048, device=cuda).to(BF)
    big_w = torch.randn(4096, 2048, device=cuda).to(BF)
    big_o = torch.empty(8192, 4096, dtype=BF, device=cuda)
    for (M, N, K) in [(64, 3072, 1024), (64, 1024, 4096), (80, 2048, 2048), (7, 1024, 2048), (128, 256, 8192)]:
        a = torch.randint(-2, 3, (M, K), generator=g).float().to(BF).to(cuda)
        w = torch.randint(-2, 3, (N, K), generator=g).float().to(BF).to(cuda)
        a = a * (torch.rand(M, K, generator=g) < 0.2).to(BF).to(cuda)    # keep |sum| < 256: exact in bf16
        ref = (a.float() @ w.float().t()).to(BF)
        out = torch.empty(M, N, dtype=BF, device=cuda)
        for it in range(12):
            with torch.cuda.stream(side):
                ops.gemm(big_a, big_w, big_o, ops.EPI_BIAS)
            out.zero_()
            ops.gemm(a, w, out, ops.EPI_BIAS)
            assert torch.equal(out, ref), f"M={M} N={N} K={K} iteration {it}: {(out != ref).sum().item()} wrong elements"
    torch.cuda.synchronize()


def test_gemm_tail_peeling_is_invisible(cuda):
    """(operands in {-1,0,1}, K=256: every intermediate is an integer below 256, exact in bf16 at both rounding points)
    automatic tile choice peels the under-filled last round into a second (128x128-tile) launch: results must not
    depend on it, including the periodic row maps of the patch-embedding epilogue"""
    from mj_video_amd import ops
    ops.gemm_set_tile(0)
    M, N, K = 17488, 2048, 256           # 69 x 8 = 552 tiles of 256^2 -> 2 full rounds + 40
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-1, 2, (M, K), generator=g).float().to(BF).to(cuda)
    w = torch.randint(-1, 2, (N, K), generator=g).float().to(BF).to(cuda)
    res = torch.randint(-8, 9, (M, N), generator=g).float().to(BF).to(cuda)
    x = res.clone()
    ops.gemm(a, w, x, ops.EPI_SCALE_RES, res=x)
    ref = (res.float() + (a.float() @ w.float().t())).to(BF)
    assert torch.equal(x, ref)
    # pos-emb style row maps whose period (1024 rows in, 1025 out) is not aligned with the peeled tail
    P, M2 = 1024, 69 * 1024
    a2 = torch.randint(-1, 2, (M2, K), generator=g).float().to(BF).to(cuda)
    w2 = torch.randint(-1, 2, (2048, K), generator=g).float().to(BF).to(cuda)
    table = torch.randint(-8, 9, (P + 1, 2048), generator=g).float().to(BF).to(cuda)
    out = torch.zeros(M2 + M2 // P, 2048, dtype=BF, device=cuda)
    ops.gemm(a2, w2, out, ops.EPI_SCALE_RES, res=table, res_mod=P, res_off=1, out_group=P, out_pad=1)
    ref2 = ((a2.float() @ w2.float().t()).view(-1, P, 2048) + table[1:].float()).to(BF)
    got = out.view(-1, P + 1, 2048)
    assert torch.equal(got[:, 1:], ref2) and (got[:, 0] == 0).all()
    # the vision tower's shape: M = 64 x 1025 = 256 x 256 + 64 -> the 64 ragged rows go to the 64 x 32 skinny kernel
    M3, N3 = 65600, 1024                 # 257 x 4 = 1028 tiles -> 4 full rounds + 4
    a3 = torch.randint(-1, 2, (M3, K), generator=g).float().to(BF).to(cuda)
    w3 = torch.randint(-1, 2, (N3, K), generator=g).float().to(BF).to(cuda)
    b3 = torch.randint(-2, 3, (N3,), generator=g).float().to(BF).to(cuda)
    ls3 = torch.randint(1, 3, (N3,), generator=g).float().to(BF).to(cuda)
    r3 = torch.randint(-8, 9, (M3, N3), generator=g).float().to(BF).to(cuda)
    x3 = r3.clone()
    ops.prof_reset()
    ops.prof_enable(True)
    ops.gemm(a3, w3, x3, ops.EPI_SCALE_RES, bias=b3, scale=ls3, res=x3)
    ops.prof_enable(False)
    tags = ops.prof_results()
    assert tags["gemm256_scale_res"]["launches"] == 1 and tags["gemm64_scale_res"]["launches"] == 1, tags
    ref3 = (r3.float() + ((a3.float() @ w3.float().t() + b3.float()).to(BF).float() * ls3.float()).to(BF).float()).to(BF)
    assert torch.equal(x3, ref3)


def test_attention_long_context_c4(cuda, attn_variant):
    """BASELINE.json configs[3] shape: one causal sequence of 28 837 tokens (16 frames x 7 tiles x 256 + text), D=128 GQA.
    The eager reference would need a 28.8k x 28.8k score matrix per head; the fp32 reference here is evaluated in query
    chunks for the first/last/middle rows of two heads."""
    from mj_video_amd import ops
    D, H, G, L = 128, 2, 2, 28837
    if two_wave_form_refused(cuda, attn_variant, D):
        return
    q, k, v = rnd(L, H * D, seed=1), rnd(L, D, seed=2), rnd(L, D, seed=3)
    cu = torch.tensor([0, L], dtype=torch.int32)
    out = torch.empty(L, H * D, dtype=BF, device=cuda)
    scale = D ** -0.5
    ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu.to(cuda), L, H, G, D, True, scale, 1)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    rows = torch.cat([torch.arange(0, 70), torch.arange(14000, 14130), torch.arange(L - 200, L)])
    kf, vf = k.float(), v.float()
    for h in range(H):
        sc = q[rows, h * D:(h + 1) * D].float() @ kf.t()
        sc = (sc.to(BF).float() * scale).to(BF).float()
        sc = sc.masked_fill(torch.arange(L)[None, :] > rows[:, None], float("-inf"))
        ref = (torch.softmax(sc, -1).to(BF).float() @ vf).to(BF).float()
        got = o[rows, h * D:(h + 1) * D]
        rel = (got - ref).norm() / ref.norm()
        assert rel.item() < 6e-3, (h, rel.item())
        assert (got - ref).abs().max().item() < 0.03


@pytest.mark.parametrize("D,H,G", [(128, 4, 2), (64, 2, 1), (96, 4, 1)])
@pytest.mark.parametrize("P,lens,qlens", [
    (0, [2186, 700, 131], [5, 700, 64]),          # suffix queries only (the last decoder layer of a scorer)
    (64, [2122, 636, 67, 1], [2122, 636, 67, 1]),  # a shared 64-key prefix, every own row a query (the prompt-prefix cache)
    (128, [2058, 300, 65], [5, 1, 33]),            # both
    (64, [700, 64], [700, 3]),
])
def test_attention_suffix_queries_and_shared_prefix(cuda, D, H, G, P, lens, qlens):
    """ABI 6 (causal): queries = the LAST rows of a sequence (cu_seqlens_q), keys = a SHARED prefix (prefix_k / prefix_v, a
    multiple of 64 rows) followed by the sequence's own rows.  Reference = the ordinary launch of the same kernel over the
    concatenated rows [prefix | own] of every sequence: include/mjv.h promises the same arithmetic with the same key-tile
    boundaries, so the selected rows must agree BIT FOR BIT (both score modes); and that ordinary launch is itself held to the
    fp32 reference by test_attention.  Older kernels and non-causal launches refuse the fields."""
    from mj_video_amd import ops, _lib
    KVH = H // G
    g = torch.Generator().manual_seed(17 + P)
    pk, pv = rnd(max(P, 1), KVH * D, seed=5)[:P], rnd(max(P, 1), KVH * D, seed=6)[:P]
    n_own = sum(lens)
    k_own, v_own = rnd(n_own, KVH * D, seed=7), rnd(n_own, KVH * D, seed=8)
    q_own = rnd(n_own, H * D, seed=9)          # a query for every own row; the extended launch uses the last qlens[i] of each
    q_pre = rnd(max(P, 1), H * D, seed=10)[:P]
    # the concatenated problem
    ks, vs, qs, full_lens = [], [], [], []
    o = 0
    for L in lens:
        ks += [pk, k_own[o:o + L]]
        vs += [pv, v_own[o:o + L]]
        qs += [q_pre, q_own[o:o + L]]
        full_lens.append(P + L)
        o += L
    kf, vf, qf = torch.cat(ks).to(cuda), torch.cat(vs).to(cuda), torch.cat(qs).to(cuda)
    cu_full = torch.tensor([0] + list(np.cumsum(full_lens)), dtype=torch.int32, device=cuda)
    cu_k = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=cuda)
    cu_q = torch.tensor([0] + list(np.cumsum(qlens)), dtype=torch.int32, device=cuda)
    q_sel = torch.cat([q_own[o + L - lq:o + L] for o, L, lq in zip(np.cumsum([0] + lens[:-1]), lens, qlens)]).to(cuda)
    scale = D ** -0.5
    for mode in (2, 1):
        full = torch.empty(qf.shape[0], H * D, dtype=BF, device=cuda)
        ops.attention(qf, kf, vf, full, cu_full, max(full_lens), H, G, D, True, scale, mode)
        want = torch.cat([full[int(cu_full[i + 1]) - lq:int(cu_full[i + 1])] for i, lq in enumerate(qlens)])
        got = torch.full((sum(qlens), H * D), float("nan"), dtype=BF, device=cuda)
        kw = {}
        if qlens != lens:
            kw.update(cu_seqlens_q=cu_q, max_seqlen_q=max(qlens))
        if P:
            kw.update(prefix_k=pk.to(cuda), prefix_v=pv.to(cuda))
        ops.attention(q_sel, k_own.to(cuda), v_own.to(cuda), got, cu_k, max(lens), H, G, D, True, scale, mode, **kw)
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, want), (mode, (got.float() - want.float()).abs().max().item())
    if kw:
        for bad in (dict(kernel=5), dict(causal=False)):
            with pytest.raises(_lib.MjvLibraryError, match="causal launches of the round-3 kernel"):
                ops.attention(q_sel, k_own.to(cuda), v_own.to(cuda), got, cu_k, max(lens), H, G, D, bad.get("causal", True), scale, 1,
                              kernel=bad.get("kernel"), **kw)
    if P:
        with pytest.raises(_lib.MjvLibraryError, match="multiple of 64"):
            ops.attention(q_sel, k_own.to(cuda), v_own.to(cuda), got, cu_k, max(lens), H, G, D, True, scale, 2,
                          prefix_k=pk[:P - 1].to(cuda), prefix_v=pv[:P - 1].to(cuda), **{k: v for k, v in kw.items() if k.startswith(("cu", "max"))})


# ------------------------------------------------------------------------------------------ event profiler
def test_event_profiler_tags_and_filter(cuda):
    """bench.py's roofline numbers come from the library's opt-in event profiler: every launch is tagged like the kernel
    rocprofv3 reports, flops are the algorithmic 2MNK, and a filter restricts recording to one tag (the timed region
    carries events for the dominant kernel only)"""
    from mj_video_amd import ops
    a, w = rnd(1024, 256, seed=1).to(cuda), rnd(512, 256, std=0.05, seed=2).to(cuda)
    out = torch.empty(1024, 512, dtype=BF, device=cuda)
    x, g, b = rnd(64, 256, seed=3).to(cuda), rnd(256, seed=4).to(cuda), rnd(256, seed=5).to(cuda)
    y = torch.empty(64, 256, dtype=BF, device=cuda)
    try:
        ops.prof_filter(None)
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(3):
            ops.gemm(a, w, out, ops.EPI_BIAS)
        ops.layernorm(x, g, b, y, 1e-6)
        ops.prof_enable(False)
        res = ops.prof_results()
        assert res["gemm256_bias"]["launches"] == 3 and res["layernorm"]["launches"] == 1
        assert res["gemm256_bias"]["flops"] == 3 * 2.0 * 1024 * 512 * 256 and res["gemm256_bias"]["ms"] > 0
        ops.prof_reset()
        ops.prof_filter("layernorm")
        ops.prof_enable(True)
        ops.gemm(a, w, out, ops.EPI_BIAS)
        ops.layernorm(x, g, b, y, 1e-6)
        ops.prof_enable(False)
        res = ops.prof_results()
        assert set(res) == {"layernorm"} and res["layernorm"]["launches"] == 1
    finally:
        ops.prof_enable(False)
        ops.prof_filter(None)
        ops.prof_reset()


def test_randomised_sweep_of_gemm_and_attention(cuda):
    """tools/fuzz_kernels.py for 20 s with a fixed seed: random GEMM problems (sizes around every tile boundary, all epilogues,
    forced and automatic tile kernels, padded row strides, with and without a split-K workspace; exact on integer data) and random
    varlen attention batches (lengths around the tile / block boundaries, both head sizes, causal or not, GQA, every kernel
    choice) against fp32 references computed with torch on the GPU.  (Round 3 ran it for 300 s: 507 000 cases, no kernel
    failure.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_kernels.py"), "20", "7"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]



# ------------------------------------------------------------------------------------------- norm folded into the GEMM
def _pad256(v, cuda):
    out = torch.zeros((v.numel() + 255) // 256 * 256, dtype=torch.float32, device=cuda)
    out[:v.numel()] = v.to(cuda)
    return out


def test_row_stats(cuda):
    from mj_video_amd import ops
    for rows, dim in ((130, 128), (1025, 1024), (333, 2048), (70, 4096)):
        x = (rnd(rows, dim, std=2.0, seed=3).float() + 0.7).to(BF)
        rstd = torch.empty(ops.padded_rows(rows), dtype=torch.float32, device=cuda)
        mrs = torch.empty_like(rstd)
        ops.row_stats(x.to(cuda), rstd, mrs, 1e-6)
        xd = x.double()
        mean = xd.mean(1)
        r = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-6)
        assert torch.allclose(rstd[:rows].cpu().double(), r, rtol=2e-6, atol=0)
        assert torch.allclose(mrs[:rows].cpu().double(), mean * r, rtol=2e-5, atol=1e-6)
        ops.row_stats(x.to(cuda), rstd, None, 1e-5)
        r2 = 1.0 / torch.sqrt((xd * xd).mean(1) + 1e-5)
        assert torch.allclose(rstd[:rows].cpu().double(), r2, rtol=2e-6, atol=0)


@pytest.mark.parametrize("M,N,K", [(515, 512, 256), (2200, 1024, 1024), (64, 256, 128)])
def test_gemm_folded_layernorm(cuda, tile, M, N, K):
    """Linear(LayerNorm(x)) with the norm folded into the GEMM (include/mjv.h "row_scale"): the kernel's Linear value is
    rstd (x W'^T - mean colsum(W')) + bias' with W' = bf16(W gamma) - against that formula in fp64 (<= 1 bf16 ulp, GELU as
    the unfused GELU test), and against the unfused composition LayerNorm -> Linear (different rounding points: relative L2)."""
    from mj_video_amd import ops
    x = (rnd(M, K, std=1.5, seed=1).float() + 0.4).to(BF)
    gamma, beta = (rnd(K, std=0.2, seed=2).float() + 1.0).to(BF), rnd(K, std=0.2, seed=3)
    w, b = rnd(N, K, std=0.05, seed=4), rnd(N, std=0.1, seed=5)
    wf = (w.float() * gamma.float()[None, :]).to(BF)
    colsum = wf.float().sum(1)
    bias2 = w.float() @ beta.float() + b.float()
    rstd = torch.empty(ops.padded_rows(M), dtype=torch.float32, device=cuda)
    mrs = torch.empty_like(rstd)
    xc = x.to(cuda)
    ops.row_stats(xc, rstd, mrs, 1e-6)
    fold = (rstd, mrs, _pad256(colsum, cuda), _pad256(bias2, cuda))
    xd = x.double()
    mean = xd.mean(1, keepdim=True)
    r = 1.0 / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-6)
    lin = (r * (xd @ wf.double().t() - mean * colsum.double()[None, :]) + bias2.double()[None, :]).float()
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(xc, wf.to(cuda), out, ops.EPI_BIAS, folded_norm=fold)
    # (the mean term cancels against the product: absolute floor = 2^-20 of the cancelling magnitudes)
    atol = float((r * (xd.abs() @ wf.double().abs().t())).max()) * 2.0 ** -20
    assert_close_bf16(out, lin.to(BF), 1, frac_exact=0.97, atol=atol, what="folded layernorm, bias")
    ops.gemm(xc, wf.to(cuda), out, ops.EPI_BIAS_GELU, folded_norm=fold)
    assert_close_bf16(out, F.gelu(lin.to(BF)), 3, frac_exact=0.96, atol=2e-3, what="folded layernorm, gelu")
    # the unfused composition the reference computes: close, not equal (the normalised rows are never rounded here)
    h = torch.empty(M, K, dtype=BF, device=cuda)
    ops.layernorm(xc, gamma.to(cuda), beta.to(cuda), h, 1e-6)
    ref = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(h, w.to(cuda), ref, ops.EPI_BIAS, bias=b.to(cuda))
    ops.gemm(xc, wf.to(cuda), out, ops.EPI_BIAS, folded_norm=fold)
    rel = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
    assert rel < 6e-3, rel


@pytest.mark.parametrize("M,K", [(515, 256), (2200, 2048), (80, 128)])
def test_gemm_folded_rmsnorm(cuda, tile, M, K):
    """Linear(RMSNorm(x)) folded: lin = rstd * (x W'^T), W' = bf16(W gain); SiLU-mul on interleaved w1 | w3 and the wqkv
    epilogue (RoPE + GQA split; on the small tiles: plain Linear + rope_split) against fp64 of that formula"""
    from mj_video_amd import ops
    x = rnd(M, K, std=1.5, seed=1)
    gain = (rnd(K, std=0.2, seed=2).float() + 1.0).to(BF)
    xc = x.to(cuda)
    rstd = torch.empty(ops.padded_rows(M), dtype=torch.float32, device=cuda)
    ops.row_stats(xc, rstd, None, 1e-5)
    xd = x.double()
    r = 1.0 / torch.sqrt((xd * xd).mean(1, keepdim=True) + 1e-5)
    ff = 256
    # (unit-scale gate / up values: a gate of -5.7 on a bf16 rounding boundary flips by one ulp under any re-association and SiLU's
    # negative tail turns that into 5 ulps of the product - seen once in 563 200 elements with sigma = 4.5)
    w1, w3 = rnd(ff, K, std=1.5 / K ** 0.5, seed=6), rnd(ff, K, std=1.5 / K ** 0.5, seed=7)
    w13 = torch.stack([w1.view(ff // 16, 16, K), w3.view(ff // 16, 16, K)], dim=1).reshape(2 * ff, K)
    w13f = (w13.float() * gain.float()[None, :]).to(BF)
    w1f, w3f = (w1.float() * gain.float()[None, :]).to(BF), (w3.float() * gain.float()[None, :]).to(BF)
    g = (r * (xd @ w1f.double().t())).float().to(BF)
    u = (r * (xd @ w3f.double().t())).float().to(BF)
    ref = (F.silu(g.float()).to(BF).float() * u.float()).to(BF)
    out = torch.empty(M, ff, dtype=BF, device=cuda)
    ops.gemm(xc, w13f.to(cuda), out, ops.EPI_SILU_MUL, folded_norm=(rstd,))
    assert_close_bf16(out, ref, 3, frac_exact=0.94, atol=2e-3, what="folded rmsnorm, silu_mul")
    # wqkv: 1 kv group of (2 q heads + k + v) x 128 columns
    G, N = 2, 512
    wq = rnd(N, K, std=1.5 / K ** 0.5, seed=8)
    wqf = (wq.float() * gain.float()[None, :]).to(BF)
    lin = (r * (xd @ wqf.double().t())).float().to(BF)
    pos = torch.arange(M, dtype=torch.int32) % 97
    half = torch.arange(0, 128, 2).float() / 128
    fr = torch.outer(torch.arange(100).float(), 1.0 / (10000.0 ** half))
    emb = torch.cat([fr, fr], dim=-1)
    cos, sin = emb.cos().to(BF).to(cuda).contiguous(), emb.sin().to(BF).to(cuda).contiguous()
    qkv = torch.zeros(M, N, dtype=BF, device=cuda)
    q, k = torch.empty(M, G * 128, dtype=BF, device=cuda), torch.empty(M, 128, dtype=BF, device=cuda)
    ops.gemm(xc, wqf.to(cuda), qkv, ops.EPI_ROPE_QKV, rope=(cos, sin, pos.to(cuda), q, k, G), folded_norm=(rstd,))
    q2, k2 = torch.empty_like(q), torch.empty_like(k)
    ops.rope_split(lin.to(cuda), q2, k2, cos, sin, pos.to(cuda), 1, G)
    # (a sum of 2048 products that cancels to ~1e-3 carries the fp32 summation-order difference of its 5-sigma terms)
    assert_close_bf16(qkv[:, (G + 1) * 128:], lin[:, (G + 1) * 128:], 1, frac_exact=0.97, atol=1e-3, what="folded rmsnorm, v columns")
    assert_close_bf16(q, q2, 2, frac_exact=0.95, atol=0.02, what="folded rmsnorm, rotated q")
    assert_close_bf16(k, k2, 2, frac_exact=0.95, atol=0.02, what="folded rmsnorm, rotated k")


def test_gemm_folded_norm_argument_errors(cuda):
    from mj_video_amd import ops, _lib
    x, w = rnd(300, 128).to(cuda), rnd(256, 128).to(cuda)
    out = torch.empty(300, 256, dtype=BF, device=cuda)
    rstd = torch.ones(512, dtype=torch.float32, device=cuda)
    with pytest.raises(_lib.MjvLibraryError, match="residual"):
        ops.gemm(x, w, out, ops.EPI_SCALE_RES, res=out, folded_norm=(rstd,))
    with pytest.raises(_lib.MjvLibraryError, match="no instantiation"):
        ops.gemm(x, w, out, ops.EPI_BIAS, folded_norm=(rstd,), tile=256)       # an RMSNorm into the plain-bias 256 kernel: not built
    ops.gemm(x, w, out, ops.EPI_BIAS, folded_norm=(rstd,), tile=128)           # ... the small kernels take every combination
    ref = torch.empty_like(out)
    ops.gemm(x, w, ref, ops.EPI_BIAS, tile=128)
    assert torch.equal(out, ref)                                               # row_scale = 1: the same Linear
