"""The MXFP8 FFN weight path (BASELINE configs[4]) on the MI355X, through the C ABI, against oracle/ref_fp8.py.

What is bit-exact: the quantiser (elements AND block scales, in the MFMA's lane layout), the norms with an MXFP8 output, the
GEMM on small-integer operands (every product and partial sum exact), and the MXFP8-output epilogue against {bf16 output,
then quantise}.  What carries a tolerance: the GEMM on random operands - the DEQUANTISED operands are the exact inputs of both
sides, only the fp32 summation order differs, so <= 1 bf16 ulp as for the bf16 GEMM - and the end-to-end model, which is held
to the fp8 ORACLE within the bf16 path's own kind of bound, and reported (not held) against the bf16 reference.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import bf16_ulps, build_hip_model, make_cfg
from test_kernels_gpu import assert_close_bf16, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _mx_to_host(mx):
    return mx.data.cpu().numpy(), mx.scales.cpu().numpy()


def _check_quantised(mx, x_bf16, what):
    """device MXFP8 (elements + scale records) == oracle quantisation of the bf16 values ``x_bf16`` (CPU tensor)"""
    from oracle import ref_fp8
    codes, sb = ref_fp8.mx_quantize(x_bf16)
    got_codes, got_scales = _mx_to_host(mx)
    # +0 / -0 are the same element value; the hardware conversion keeps the sign of a zero, so does the oracle: compare bits
    assert np.array_equal(got_codes, codes), f"{what}: {(got_codes != codes).sum()} of {codes.size} e4m3 elements differ"
    rec = ref_fp8.mx_scale_records(sb)
    rows = x_bf16.shape[0]
    # bytes of rows beyond the matrix (last 64-row group) are never written by the kernels: compare the real rows only
    mask = ref_fp8.mx_scale_records(np.ones_like(sb)) != 0
    assert rec.size == got_scales.size or got_scales.size >= rec.size
    assert np.array_equal(got_scales[:rec.size][mask], rec[mask]), f"{what}: block scales differ (rows {rows})"


# ------------------------------------------------------------------------------------------- quantiser
@pytest.mark.parametrize("rows,cols", [(64, 128), (1, 128), (77, 256), (300, 1024), (1025, 4096), (130, 8192)])
def test_quantize_mxfp8_bit_exact(cuda, rows, cols):
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + cols)
    # magnitudes over 40 binades row by row, a few exact zeros, whole zero blocks, the block-maximum corner 1.75 / 1.76 * 2^e
    x = torch.randn(rows, cols, generator=g) * torch.logspace(-20, 20, rows, base=2.0)[:, None]
    x[torch.rand(rows, cols, generator=g) < 0.02] = 0.0
    x[0, :64] = 0.0
    if rows > 2:
        x[1, :32] = torch.tensor([1.75] + [0.01] * 31)          # amax exactly 448 / 256: no bump
        x[2, :32] = torch.tensor([1.7578125] + [0.3] * 31)      # the next bf16 above 1.75: the scale moves up one binade
    x = x.to(BF)
    mx = ops.quantize_mxfp8(x.to(cuda))
    torch.cuda.synchronize()
    _check_quantised(mx, x, f"quantize {rows}x{cols}")
    # and nothing saturates / no NaN element
    codes = mx.data.cpu().numpy()
    assert ((codes & 0x7F) != 0x7F).all()


def test_quantize_mxfp8_every_bf16_magnitude(cuda):
    """every finite bf16 value as a block maximum (block = [v, v/2, v/3, ...]): scale choice and element rounding, exhaustively"""
    from mj_video_amd import ops
    u = torch.arange(0, 0x7F80, dtype=torch.int32).to(torch.int16).view(BF).float()        # all non-negative finite bf16
    u = u[: (u.numel() // 64) * 64]
    div = torch.arange(1, 33, dtype=torch.float32)
    x = (u[:, None] / div[None, :]).to(BF)                                                # [n, 32]
    x = x.reshape(-1, 128)                                                                # 4 blocks per row
    x[::2] = -x[::2]
    mx = ops.quantize_mxfp8(x.to(cuda))
    torch.cuda.synchronize()
    _check_quantised(mx, x, "every bf16 magnitude")


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,dim", [(130, 128), (1025, 1024), (333, 2048), (70, 4096)])
def test_norms_mxfp8_equal_bf16_norm_then_quantise(cuda, rows, dim):
    from mj_video_amd import ops
    x = rnd(rows, dim, std=2.0, seed=3).to(cuda)
    g, b = rnd(dim, std=0.2, seed=4).to(cuda) + 1.0, rnd(dim, std=0.1, seed=5).to(cuda)
    g = g.to(BF)
    ref = torch.empty_like(x)
    ops.layernorm(x, g, b, ref, 1e-6)
    out = ops.MX8.empty(rows, dim, cuda)
    ops.layernorm_mxfp8(x, g, b, out, 1e-6)
    torch.cuda.synchronize()
    _check_quantised(out, ref.cpu(), "layernorm_mxfp8")
    ops.rmsnorm(x, g, ref, 1e-5)
    ops.rmsnorm_mxfp8(x, g, out, 1e-5)
    torch.cuda.synchronize()
    _check_quantised(out, ref.cpu(), "rmsnorm_mxfp8")


# ------------------------------------------------------------------------------------------- GEMM
# Measured (tests/diag_mfma_fp8_accumulation.py, profiles/r04_c_mfma_fp8_accumulation.txt): the sum of the 128 products INSIDE one
# v_mfma_scale_f32_16x16x128_f8f6f4 is not a sequential fp32 sum - on random operands it is off by up to 2^-17 of sum|a||w|
# (K = 128; 2^-19 at K = 1024), against 2^-24.7 for torch's fp32 matmul.  An output with heavy cancellation (|sum| << sum|a||w|)
# is therefore several bf16 ulps of ITSELF away from the exact sum although every product is exact.  The tolerance of the
# random-operand tests is stated accordingly: one bf16 ulp of the exact result + 2^-14 * sum|a||w|.  (The 2^-17 is the maximum over
# 65 k outputs; the distribution has a tail: over 5e8 outputs at K = 128 the maximum is 2^-14.9, 83 beyond 2^-16, ONE beyond 2^-15 -
# tests/diag_mfma_fp8_accumulation_tail.py, profiles/r04_u_mfma_fp8_accumulation_tail.txt, found after a 1.6-million-case fuzz run reported
# one output of 3.7e11 beyond the old 2^-15; per output the excess is 2^-18.6 x 128 x its largest |a w| product, what an adder
# aligned to the largest term would do.)
ACC_TOL = 2.0 ** -14


def _abs_products(aq, wq):
    return (aq.double().abs() @ wq.double().abs().t())


def _quant_pair(a, w, cuda):
    from mj_video_amd import ops
    from oracle import ref_fp8
    a8, w8 = ops.quantize_mxfp8(a.to(cuda)), ops.quantize_mxfp8(w.to(cuda))
    return a8, w8, ref_fp8.mx_fake_quant(a), ref_fp8.mx_fake_quant(w)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (200, 136, 256), (513, 520, 128), (700, 256, 384), (256, 1024, 640), (2200, 768, 2048), (64, 8, 128)])
def test_gemm_mxfp8_exact_integers(cuda, M, N, K):
    """small-integer operands quantise exactly (x * 2^5 <= 256 has <= 4 significant bits) and every partial sum is exact in
    fp32: the result must be bit-exact - catches any operand-map / scale-record / swizzle / pipeline-race mistake; K = 1, 2,
    3, 5, 16 K-tiles of the pipeline and of the 4-slot scale ring, M and N ragged against the tile; per-row and per-block
    power-of-two factors exercise the scales (products stay exact)"""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-8, 9, (M, K), generator=g).float()
    w = torch.randint(-7, 8, (N, K), generator=g).float()
    # (exponents in [-1, 1]: |sum| < 56 * 4 * 2048 < 2^19 in units of 2^-2 - 21 bits, exact in an fp32 accumulator)
    a = a * torch.exp2(torch.randint(-1, 2, (M, K // 32), generator=g).float()).repeat_interleave(32, dim=1)
    w = w * torch.exp2(torch.randint(-1, 2, (N, K // 32), generator=g).float()).repeat_interleave(32, dim=1)
    a, w = a.to(BF), w.to(BF)
    a8, w8, aq, wq = _quant_pair(a, w, cuda)
    assert torch.equal(aq, a.float()) and torch.equal(wq, w.float()), "the integer operands must quantise exactly"
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a8, w8, out, ops.EPI_BIAS)
    ref = (a.double() @ w.double().t()).float().to(BF)
    assert torch.equal(out.cpu(), ref), f"{(out.cpu() != ref).sum().item()} of {ref.numel()} differ"


@pytest.mark.parametrize("out8", [False, True])
def test_gemm_mxfp8_gelu_epilogue_every_bf16_value(cuda, out8):
    """the fp8 kernel's GELU epilogue (round 5: the split-table form of the bf16 kernel - gather address = the value's bit pattern,
    one wave-wide range vote) on EVERY bf16 value, NaNs and infinities included: A = 0, so a pre-activation is its column's bias
    exactly; 256 values per launch = 16 votes of 16 columns, natural order (whole votes inside / outside the table) and
    shuffled (mixed votes -> the general path).  Bit for bit torch's CPU bf16 GELU (NaN where it gives NaN); the MXFP8 output
    equals the quantised bf16 output."""
    from mj_video_amd import ops
    M, N, K = 256, 256, 128
    a8 = ops.quantize_mxfp8(torch.zeros(M, K, dtype=BF, device=cuda))
    w8 = ops.quantize_mxfp8(rnd(N, K, seed=2).to(cuda))
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(BF)
    for order in ("natural", "shuffled"):
        v = bits if order == "natural" else bits[torch.randperm(65536, generator=torch.Generator().manual_seed(11))]
        for c0 in range(0, 65536, N):
            b = v[c0:c0 + N]
            out = torch.empty(M, N, dtype=BF, device=cuda)
            ops.gemm(a8, w8, out, ops.EPI_BIAS_GELU, bias=b.to(cuda))
            ref = F.gelu(b)
            got = out[[0, 77, 255]].cpu()
            want = ref.expand(3, N)
            nan = torch.isnan(want)
            assert torch.equal(torch.isnan(got), nan), (order, c0)
            same = got.view(torch.int16) == want.view(torch.int16)
            same |= (got.float() == 0) & (want.float() == 0)
            same |= b.float().abs().expand(3, N) < 2.0 ** -125       # (torch flushes subnormal results, the table's x / 2 keeps them)
            assert (same | nan).all(), (order, c0, b[~(same | nan)[0]][:8], got[0][~(same | nan)[0]][:8], ref[~(same | nan)[0]][:8])
            if out8 and not nan.any() and torch.isfinite(ref.float()).all():
                o8 = ops.MX8.empty(M, N, cuda)
                ops.gemm(a8, w8, o8, ops.EPI_BIAS_GELU, bias=b.to(cuda))
                chk = ops.quantize_mxfp8(out)
                assert torch.equal(o8.data, chk.data) and torch.equal(o8.scales, chk.scales), (order, c0)


@pytest.mark.parametrize("M,N,K", [(2112, 8192, 1024),    # 9 x 32 tiles on 256 CUs: a 64-row tail peeled off and sliced 4 x
                                   (2304 + 80, 7168, 2048),  # 10 x 28 = 280 tiles: the last m-tile (80 rows) peeled, sliced 8 x
                                   (200, 512, 4096),         # under-filled as a whole (2 tiles): sliced 16 x
                                   (77, 1024, 512)])         # 4 tiles, 4 K-tiles: too short to slice - the plain launch
@pytest.mark.parametrize("epi,out8", [("bias", False), ("bias", True), ("gelu", False), ("gelu", True), ("scale_res", False),
                                      ("silu", False), ("silu", True)])     # (the residual epilogue has no MXFP8 output)
def test_gemm_mxfp8_sliced_tails_equal_the_unsliced_launch(cuda, M, N, K, epi, out8):
    """round 5: with a workspace, the m-tile rows of an under-filled last round (and under-filled launches as a whole) run
    K-sliced and are finished by splitk_finish256f8_kernel - its own epilogue code, its own block quantiser.  On small-integer
    operands every partial sum is exact, so the sliced result must equal the unsliced launch (no workspace) BIT FOR BIT: bf16
    outputs, MXFP8 elements and scale records, every epilogue."""
    from mj_video_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-8, 9, (M, K), generator=g).float()
    w = torch.randint(-7, 8, (N, K), generator=g).float()
    a = a * torch.exp2(torch.randint(-1, 2, (M, K // 32), generator=g).float()).repeat_interleave(32, dim=1)
    w = w * torch.exp2(torch.randint(-1, 2, (N, K // 32), generator=g).float()).repeat_interleave(32, dim=1)
    # keep the sums small enough for GELU / SiLU to see their interesting range: scale the weights down by a power of two
    w = w * 2.0 ** -8
    a8, w8 = ops.quantize_mxfp8(a.to(BF).to(cuda)), ops.quantize_mxfp8(w.to(BF).to(cuda))
    e = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "scale_res": ops.EPI_SCALE_RES, "silu": ops.EPI_SILU_MUL}[epi]
    nout = N // 2 if epi == "silu" else N
    bias = None if epi == "silu" else (torch.randint(-4, 5, (N,), generator=g).float() * 0.25).to(BF).to(cuda)
    scale = (torch.randint(1, 4, (N,), generator=g).float() * 0.5).to(BF).to(cuda) if epi == "scale_res" else None
    res = torch.randint(-8, 9, (M, N), generator=g).float().to(BF).to(cuda) if epi == "scale_res" else None
    ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=cuda)

    def run(workspace):
        out = ops.MX8.empty(M, nout, cuda) if out8 else torch.full((M, nout), float("nan"), dtype=BF, device=cuda)
        if out8:
            out.data.fill_(0x7F)
            out.scales.zero_()
        ops.gemm(a8, w8, out, e, bias=bias, scale=scale, res=(res.clone() if res is not None else None), workspace=workspace, tile=0)
        return out

    ops.prof_filter(None); ops.prof_reset(); ops.prof_enable(True)
    try:
        sliced = run(ws)
        torch.cuda.synchronize()
    finally:
        ops.prof_enable(False)
    tags = ops.prof_results()
    ops.prof_reset()
    plain = run(None)
    torch.cuda.synchronize()
    if (M, N, K) != (77, 1024, 512):
        assert any(t.startswith("gemm256f8s_") for t in tags), tags.keys()    # the sliced kernels really ran
    else:
        assert not any(t.startswith("gemm256f8s_") for t in tags), tags.keys()
    if out8:
        assert torch.equal(sliced.data, plain.data), int((sliced.data != plain.data).sum())
        assert torch.equal(sliced.scales, plain.scales), int((sliced.scales != plain.scales).sum())
        assert not (plain.data == 0x7F).all(dim=1).any()
    else:
        assert torch.isfinite(plain.float()).all()
        assert torch.equal(sliced, plain), int((sliced != plain).sum())


@pytest.mark.parametrize("M,N,K", [(300, 384, 128), (1025, 3072, 1024), (77, 512, 256), (640, 1024, 4096)])
def test_gemm_mxfp8_bias_random(cuda, M, N, K):
    from mj_video_amd import ops
    a, w, b = rnd(M, K, seed=1), rnd(N, K, std=0.05, seed=2), rnd(N, std=0.1, seed=3)
    a8, w8, aq, wq = _quant_pair(a, w, cuda)
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a8, w8, out, ops.EPI_BIAS, bias=b.to(cuda))
    exact = aq.double() @ wq.double().t() + b.double()
    ref = exact.float().to(BF)
    T = _abs_products(aq, wq)
    err = (out.double().cpu() - exact).abs()
    ok = err <= exact.abs() * 2.0 ** -8 + ACC_TOL * T
    assert ok.all(), f"{int((~ok).sum())} of {ok.numel()} outputs beyond 1 bf16 ulp + 2^-14 sum|a||w| (worst excess / T: {((err - exact.abs() * 2.0 ** -8) / T).max().item():.2e})"
    assert (bf16_ulps(out.float().cpu(), ref.float()) == 0).float().mean().item() >= 0.97
    # how far the fp8 product is from the bf16 product of the unquantised operands: reported, loosely bounded (2^-4 operands)
    full = (a.double() @ w.double().t() + b.double()).float()
    rel = ((out.float().cpu() - full).norm() / full.norm()).item()
    assert rel < 0.06, rel


def test_gemm_mxfp8_epilogues(cuda):
    """GELU / ReLU / LayerScale+residual / SiLU-mul epilogues on MXFP8 operands = the bf16 kernels' epilogue code on the fp8
    accumulators: against the same torch composition as tests/test_kernels_gpu.py, fed the dequantised operands"""
    from mj_video_amd import ops
    M, N, K = 515, 512, 256
    a, w, b = rnd(M, K, seed=1), rnd(N, K, std=0.1, seed=2), rnd(N, std=0.1, seed=3)
    a8, w8, aq, wq = _quant_pair(a, w, cuda)
    lin = (aq.double() @ wq.double().t() + b.double()).float().to(BF)
    atol = ACC_TOL * _abs_products(aq, wq).max().item()     # (see ACC_TOL: near-zero outputs of a cancelling sum)
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(a8, w8, out, ops.EPI_BIAS_GELU, bias=b.to(cuda))
    # (as test_kernels_gpu.py::test_gemm_gelu_relu: GELU's negative tail amplifies a 1-ulp flip of the Linear - relative
    # sensitivity ~ -9 at x = -3 - on outputs below 1e-2: absolute floor 2e-3)
    # and where x in [2, 2.05) maps to gelu(x) in [1.95, 2) one ulp of x is 2.2 ulps of the result: 3 ulps
    assert_close_bf16(out, F.gelu(lin), 3, frac_exact=0.96, atol=max(atol, 2e-3), what="mxfp8 gelu")
    ops.gemm(a8, w8, out, ops.EPI_BIAS_RELU, bias=b.to(cuda))
    assert_close_bf16(out, F.relu(lin), 2, frac_exact=0.97, atol=atol, what="mxfp8 relu")
    ls, res = rnd(N, std=0.3, seed=4), rnd(M, N, seed=5)
    ops.gemm(a8, w8, out, ops.EPI_SCALE_RES, bias=b.to(cuda), scale=ls.to(cuda), res=res.to(cuda))
    ref = (res.float() + (lin.float() * ls.float()).to(BF).float()).to(BF)
    # (as the bf16 test: a 1-ulp flip of the rounded GEMM term, |v| up to ~4 -> 2^-6, survives cancellation against the residual)
    assert_close_bf16(out, ref, 2, frac_exact=0.95, atol=0.02, what="mxfp8 scale_res")
    # in place on the residual stream, as the model calls it
    x = res.to(cuda).clone()
    ops.gemm(a8, w8, x, ops.EPI_SCALE_RES, bias=b.to(cuda), scale=ls.to(cuda), res=x)
    assert torch.equal(x, out)
    # SiLU-mul on 16-row interleaved w1 | w3
    ff = 256
    w1, w3 = rnd(ff, K, std=0.1, seed=6), rnd(ff, K, std=0.1, seed=7)
    w13 = torch.stack([w1.view(ff // 16, 16, K), w3.view(ff // 16, 16, K)], dim=1).reshape(2 * ff, K).contiguous()
    w13_8 = ops.quantize_mxfp8(w13.to(cuda))
    from oracle import ref_fp8
    g = (aq.double() @ ref_fp8.mx_fake_quant(w1).double().t()).float().to(BF)
    u = (aq.double() @ ref_fp8.mx_fake_quant(w3).double().t()).float().to(BF)
    ref = (F.silu(g.float()).to(BF).float() * u.float()).to(BF)
    o2 = torch.empty(M, ff, dtype=BF, device=cuda)
    ops.gemm(a8, w13_8, o2, ops.EPI_SILU_MUL)
    assert_close_bf16(o2, ref, 3, frac_exact=0.92, atol=max(atol, 2e-3), what="mxfp8 silu_mul")


@pytest.mark.parametrize("epi", ["bias", "gelu", "relu", "silu"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (515, 1024, 256), (1100, 512, 1024)])
def test_gemm_mxfp8_output_equals_bf16_output_then_quantise(cuda, epi, M, N, K):
    """c_format MXFP8: bit-identical (elements and scale records) to the same GEMM with a bf16 output followed by the
    quantiser - the fused epilogue changes where the bytes go, not what they are"""
    from mj_video_amd import ops
    code = dict(bias=ops.EPI_BIAS, gelu=ops.EPI_BIAS_GELU, relu=ops.EPI_BIAS_RELU, silu=ops.EPI_SILU_MUL)[epi]
    a, w = rnd(M, K, seed=11), rnd(N, K, std=0.08, seed=12)
    b = None if epi == "silu" else rnd(N, std=0.1, seed=13).to(cuda)
    a8, w8 = ops.quantize_mxfp8(a.to(cuda)), ops.quantize_mxfp8(w.to(cuda))
    nout = N // 2 if epi == "silu" else N
    ref = torch.empty(M, nout, dtype=BF, device=cuda)
    ops.gemm(a8, w8, ref, code, bias=b)
    out = ops.MX8.empty(M, nout, cuda)
    out.data.fill_(0x55)
    ops.gemm(a8, w8, out, code, bias=b)
    torch.cuda.synchronize()
    _check_quantised(out, ref.cpu(), f"mxfp8 output, {epi}")


def test_gemm_mxfp8_repeat_launch_stability(cuda):
    """race screen of the new pipeline pieces (scale ring, 5-instruction counted waits): 200 back-to-back launches of three
    shapes on a busy chip, every result bit-equal to the first"""
    from mj_video_amd import ops
    for M, N, K in ((1024, 1024, 512), (4096, 2048, 2048), (777, 512, 8192)):
        a8, w8 = ops.quantize_mxfp8(rnd(M, K, seed=21).to(cuda)), ops.quantize_mxfp8(rnd(N, K, std=0.05, seed=22).to(cuda))
        first = torch.empty(M, N, dtype=BF, device=cuda)
        ops.gemm(a8, w8, first, ops.EPI_BIAS)
        out = torch.empty_like(first)
        bad = 0
        for _ in range(200):
            ops.gemm(a8, w8, out, ops.EPI_BIAS)
            bad += int(not torch.equal(out, first))
        assert bad == 0, f"{M}x{N}x{K}: {bad} of 200 launches differ from the first"


def test_gemm_mxfp8_argument_errors(cuda):
    from mj_video_amd import ops, _lib
    a8, w8 = ops.quantize_mxfp8(rnd(64, 128).to(cuda)), ops.quantize_mxfp8(rnd(64, 128).to(cuda))
    out = torch.empty(64, 64, dtype=BF, device=cuda)
    with pytest.raises(_lib.MjvLibraryError, match="residual"):
        ops.gemm(a8, w8, out, ops.EPI_SCALE_RES)   # no residual given
    with pytest.raises(AssertionError):
        ops.gemm(a8, rnd(64, 128).to(cuda), out, ops.EPI_BIAS)                              # mixed operand formats


# ------------------------------------------------------------------------------------------- model
def _tiny_model_and_batch(cuda, seed=21):
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    cfg = make_cfg("tiny", 56)
    sd = synth.synth_state_dict(cfg, seed=seed)                      # bf16 tensors, checkpoint layout
    model = build_hip_model(cfg, sd, cuda)
    per = num_image_tokens_per_tile(cfg)
    px = torch.cat([synth.synth_pixel_values(9, 0, 4, 56), synth.synth_pixel_values(9, 1, 3, 56)])
    ids, mask = synth.pad_batch([synth.synth_input_ids(4 * per, 1), synth.synth_input_ids(3 * per, 2)])
    return cfg, sd, model, px, ids, mask


def test_model_mxfp8_against_fp8_oracle_tiny(cuda):
    """the whole forward with the mxfp8 FFN path against oracle/ref_fp8.py (right-padded batch of two, tiny dims - every FFN
    K is 128, 256 or 512): each CustomOutput field as close to the fp8 oracle as the bf16 path is to the bf16 oracle on the same
    inputs (x 3 + a floor: the kernels around the FFN and the accumulation-order noise are the same; an activation that lands
    on the other side of an e4m3 rounding boundary moves by 2^-4 of itself, not 2^-9), and the fp8-vs-bf16 distance reported"""
    from mj_video_amd import synth
    from oracle import ref_cpu, ref_fp8
    cfg, sd, model, px, ids, mask = _tiny_model_and_batch(cuda)
    model.set_ffn_format("mxfp8")
    out8 = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    model.set_ffn_format("bf16")
    out16 = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    ref8 = ref_fp8.reward_forward_fp8(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    ref16 = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    worst = {}
    for k in ("score", "aspect_scores", "rewards", "aspect_gating_output", "aspect_weights", "criteria_gating_output", "hidden_state"):
        d8 = (getattr(out8, k).float().cpu() - ref8[k].float()).abs().max().item()
        d16 = (getattr(out16, k).float().cpu() - ref16[k].float()).abs().max().item()
        gap = (ref8[k].float() - ref16[k].float()).abs().max().item()
        worst[k] = (round(d8, 5), round(d16, 5), round(gap, 5))
        # (the allowance beyond 3 x the bf16 path's own distance is DERIVED, not fitted to a sample: two fp8 implementations
        # that sum in different orders may disagree by up to the distance between the fp8 and the bf16 ORACLE on this very input -
        # an activation that falls on the other side of an e4m3 rounding boundary is one of the moves that make up that gap.
        # Measured in round 5: rewards 0.0642 against a gap of 0.0742; the values of every run are in the message)
        assert d8 <= 3.0 * d16 + 1.0 * gap + 2e-3, (k, dict(hip8_vs_oracle8=d8, hip16_vs_oracle16=d16, oracle8_vs_oracle16=gap))
    print("tiny: |hip8 - oracle8|, |hip16 - oracle16|, |oracle8 - oracle16| per field:", worst)
    # switching formats changes the result (the path is really taken)
    assert worst["hidden_state"][2] > 0 and not torch.equal(out8.score, out16.score)


def test_model_mxfp8_deterministic_and_reprepares(cuda):
    cfg, sd, model, px, ids, mask = _tiny_model_and_batch(cuda, seed=22)
    a = model.set_ffn_format("mxfp8").forward(px.to(cuda), ids.to(cuda), mask.to(cuda)).score.clone()
    b = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda)).score.clone()
    assert torch.equal(a, b)
    c = model.set_ffn_format("bf16").forward(px.to(cuda), ids.to(cuda), mask.to(cuda)).score.clone()
    d = model.set_ffn_format("mxfp8").forward(px.to(cuda), ids.to(cuda), mask.to(cuda)).score.clone()
    assert torch.equal(a, d) and not torch.equal(a, c)
    with pytest.raises(ValueError):
        model.set_ffn_format("int4")


def test_attention_side_mxfp8_study_flag(cuda):
    """``model._exp_fp8_attn_side`` - the measurement flag behind tools/fp8_attn_side_study.py (MXFP8 on qkv / proj, wqkv / wo too,
    through unfused launches; DESIGN "Why the fp8 path stops at the FFN") - keeps working: finite scores that differ from the
    FFN-only path, the FFN-only result bit for bit again once cleared, and a deviation from bf16 of the same order as the FFN path's"""
    cfg, sd, model, px, ids, mask = _tiny_model_and_batch(cuda, seed=23)
    run = lambda: model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda)).score.float().clone()   # noqa: E731
    bf = run()
    ffn = model.set_ffn_format("mxfp8") and run()
    model._exp_fp8_attn_side = True
    allq = run()
    model._exp_fp8_attn_side = False
    back = run()
    model.set_ffn_format("bf16")
    assert torch.isfinite(allq).all() and not torch.equal(allq, ffn) and torch.equal(back, ffn)
    assert (allq - bf).abs().max().item() <= 10.0 * max((ffn - bf).abs().max().item(), 1e-2)


@pytest.mark.parametrize("tower,layer", [("vit", 0), ("vit", 23), ("llm", 0), ("llm", 23)])
def test_single_layer_mxfp8_at_production_shape(cuda, tower, layer):
    """ONE layer at MJ-VIDEO-2B dims and the headline sequence lengths (2 x 1025 x 1024 vision rows, 1 x 2186 x 2048 language
    rows, seed-defined inputs and weights as tests/golden/make_layer_fixtures.py builds them): the HIP layer in mxfp8 mode
    against the fp8 ORACLE's layer (oracle/ref_cpu.py's layer function under ref_fp8.fp8_ffn), no compounding over 24 layers.
    Yardsticks measured in the same test: the bf16 HIP layer against the bf16 oracle layer, and the fp8-vs-bf16 gap of the
    oracle.  Bound: the fp8 HIP layer is within 4 x the bf16 distance + 0.3 % of ITS oracle - two fp8 implementations that sum in
    different orders disagree on the ~1e-3 of the FFN inputs whose bf16 value sits on an e4m3 rounding boundary, and such an
    element moves by 2^-3 of itself, not 2^-8 - and at least 3 x closer to it than the bf16 oracle is (the fp8-vs-bf16 gap)."""
    from util import layer_input_rows, layer_tensors, load_golden
    from mj_video_amd.modeling import InternVLChatRewardModeling
    from oracle import ref_cpu, ref_fp8
    npz, meta = load_golden("layers")
    m = meta["layers"]
    case = next(c for c in m["cases"] if c["name"] == f"{tower}_layer{layer}")
    cfg = make_cfg("2b", m["image_size"])
    prefix = (f"model.vision_model.encoder.layers.{layer}." if tower == "vit" else f"model.language_model.model.layers.{layer}.")
    w = layer_tensors(cfg, prefix, m["weight_seed"])
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 1
    cfg.llm_config.vocab_size = 128
    model = InternVLChatRewardModeling.from_config(cfg, dtype=BF)
    for prm in model.parameters():
        prm.data.zero_()
    mod = model.model.vision_model.encoder.layers[0] if tower == "vit" else model.model.language_model.model.layers[0]
    mod.load_state_dict(w, strict=True)
    model = model.to(BF).to(cuda).eval()
    x = layer_input_rows(m["input_seed"], case["input_tag"], tuple(case["shape"]))
    run = model.run_vit_layer if tower == "vit" else model.run_llm_layer
    y8 = run(0, x).float().cpu() if model.set_ffn_format("mxfp8") else None
    y16 = run(0, x).float().cpu() if model.set_ffn_format("bf16") else None
    # the oracle's layer on the same rows / weights (keys as ref_cpu expects them: layer index 0)
    p0 = "model.vision_model.encoder.layers.0." if tower == "vit" else "model.language_model.model.layers.0."
    sd = {p0 + k: v for k, v in w.items()}

    def oracle_layer():
        if tower == "vit":
            return ref_cpu.vit_layer(sd, cfg, 0, x)
        B, N, _ = x.shape
        mask = ref_cpu.causal_padding_mask(torch.ones(B, N, dtype=torch.bool), x.dtype)
        cos, sin = ref_cpu.rope_tables(cfg, N, x.dtype)
        return ref_cpu.llm_layer(sd, cfg, 0, x, mask, cos, sin)

    r16 = oracle_layer().float()
    with ref_fp8.fp8_ffn():
        r8 = oracle_layer().float()

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()

    d8, d16, gap = rel(y8, r8), rel(y16, r16), rel(r8, r16)
    print(f"{tower}_layer{layer} @ production shape: hip8 vs oracle8 {d8:.5f}; hip16 vs oracle16 {d16:.5f}; oracle8 vs oracle16 {gap:.5f}")
    assert torch.isfinite(y8).all()
    # measured (profiles/r04_d_fp8_parity.txt): vision 0.47 % against 0.18 % (bf16) and a 2.4 % gap; language 1.1 % / 0.29 % / 4.3 %
    # (4 x: the language layer measures 3.8 - 4.0 x under either attention numerics)
    assert d8 <= 4.0 * d16 + 3e-3, (d8, d16)
    assert d8 * 3.0 <= gap, (d8, gap)


def test_model_mxfp8_against_fp8_oracle_c1_dims(cuda):
    """MJ-VIDEO-2B dims, one video of 8 tiles @224^2 (BASELINE configs[0]'s shape), mxfp8 FFN path against the fp8 ORACLE
    run here on the host (about half a minute): final hidden rows and every head output.  The yardstick is measured in the same
    test: the bf16 HIP path against the bf16 oracle on the same input (the two paths share every kernel but the FFN's)."""
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    from oracle import ref_cpu, ref_fp8
    cfg = make_cfg("2b", 224)
    sd = synth.synth_state_dict(cfg, seed=5, lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=BF).expand(cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    px = synth.synth_pixel_values(77, 0, 8, 224)
    ids = synth.synth_input_ids(num_image_tokens_per_tile(cfg) * 8, caption_seed=3)
    mask = torch.ones_like(ids)
    out8 = model.set_ffn_format("mxfp8").forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    out16 = model.set_ffn_format("bf16").forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    ref8 = ref_fp8.reward_forward_fp8(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    ref16 = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)

    def rel(a, b):
        a, b = a.float().cpu(), b.float()
        return ((a - b).norm() / b.norm()).item()

    r8 = {k: rel(getattr(out8, k), ref8[k]) for k in ("hidden_state", "prompt_embedding")}
    r16 = {k: rel(getattr(out16, k), ref16[k]) for k in ("hidden_state", "prompt_embedding")}
    gap = {k: rel(ref8[k], ref16[k]) for k in ("hidden_state", "prompt_embedding")}
    print(f"2B dims @224: relative L2 of the hidden rows: hip8 vs oracle8 {r8}; hip16 vs oracle16 {r16}; oracle8 vs oracle16 {gap}")
    print(f"score: hip8 {out8.score.item():+.4f} oracle8 {ref8['score'].item():+.4f} | hip16 {out16.score.item():+.4f} oracle16 {ref16['score'].item():+.4f}")
    # End to end the statement is statistical, as for the bf16 path (DESIGN 2 "how chaotic the path is"): after 48 layers the
    # per-layer disagreement of two fp8 implementations (single-layer test above: 0.5 % / 1.1 %, about a quarter of the layer's
    # fp8-vs-bf16 gap) has compounded to about two thirds of the compounded gap (measured 12.3 % / 13.3 % against 18.5 % / 19.7 %;
    # the bf16 path: 2.3 % / 2.5 %).  The bound is what the single-layer test implies: its per-layer bound
    # (4 x the bf16 layer distance + 0.3 % = 1.0 % / 1.5 %) summed in quadrature over 24 + 24 layers is 8.7 %, times the
    # amplification the bf16 path shows between its own layer distances and its end-to-end one (2.3 % from 1.67 %: 1.38 x) = 12 %,
    # i.e. 5.2 x the bf16 end-to-end distance measured in this very test -> held to 6 x (round 4 accepted 8 x), to 0.75 of the
    # fp8-vs-bf16 gap (round 4: 1.0), and the score to 0.3 of the oracles' own fp8-vs-bf16 score gap (measured 0.20; round 4
    # accepted max(gap, 0.25), i.e. any score between the two oracles).
    for k in r8:
        assert r8[k] <= 0.75 * gap[k], (k, r8[k], gap[k])
        assert r8[k] <= 6.0 * r16[k], (k, r8[k], r16[k])
    for k in ("score", "aspect_scores", "rewards", "aspect_gating_output"):
        assert torch.isfinite(getattr(out8, k).float()).all(), k
    d_score = abs(out8.score.item() - ref8["score"].item())
    gap_score = abs(ref8["score"].item() - ref16["score"].item())
    print(f"score: |hip8 - oracle8| {d_score:.4f} = {d_score / gap_score:.3f} of the oracles' fp8-vs-bf16 gap {gap_score:.4f}")
    assert d_score <= max(0.3 * gap_score, 0.05), (d_score, gap_score)


def _rank_case_engineered_mxfp8(cuda, name, pairs_per_forward, max_noise_ratio, min_agree, min_rho, fmt="mxfp8"):
    from scipy.stats import spearmanr
    from test_e2e_gpu import _rank_run
    run = _rank_run(cuda, name, pairs_per_forward, ffn_format=fmt)
    eng_name = name.replace("rankset", "rankeng")
    if run["eng"] is None:
        pytest.skip(f"{eng_name} fixture not generated")
    from util import load_golden
    enpz, emeta = load_golden(eng_name)
    ref, keep, got = enpz["ref_bf16"], enpz["keep"], run["eng"][: enpz["ref_bf16"].shape[0]]
    f32, idx32 = enpz["ref_fp32"], enpz["fp32_pairs"]
    noise_rms = float(np.sqrt(((ref[idx32][..., 0] - f32[..., 0]) ** 2).mean()))
    d = (got[..., 0] - ref[..., 0]).ravel()
    rms = float(np.sqrt((d ** 2).mean()))
    agree = np.sign(got[:, 0, 0] - got[:, 1, 0]) == np.sign(ref[:, 0, 0] - ref[:, 1, 0])
    rho = spearmanr(got[..., 0].ravel(), ref[..., 0].ravel()).correlation
    print(f"{fmt} vs reference bf16, {eng_name}: score spread {float(ref[..., 0].std()):.4f}; reference bf16-vs-fp32 noise rms {noise_rms:.5f}; "
          f"|hip8 - ref| rms {rms:.5f} ({rms / noise_rms:.1f} x) max {np.abs(d).max():.5f}; preference agreement on the {int(keep.sum())} decisive "
          f"pairs {agree[keep].mean():.5f} ({int((~agree[keep]).sum())} flips), on all {len(agree)} pairs {agree.mean():.5f}; spearman {rho:.6f}")
    assert np.isfinite(got).all()
    assert rms <= max_noise_ratio * noise_rms, (rms, noise_rms)
    assert agree[keep].mean() >= min_agree
    assert rho >= min_rho


def test_rank_agreement_engineered_c1_mxfp8(cuda):
    """The fp8 FFN path against the REFERENCE's bf16 scores on the engineered rank set @224^2 (512 pairs, 478 decisive): reported
    with ITS OWN stated tolerance - this path is not a drop-in for the bf16 numbers (2^-4 operand rounding against 2^-9, through
    a backbone that amplifies rounding noise): score deviation rms <= 10 x the reference's own bf16-vs-fp32 noise (CPU study,
    DESIGN 7.4: 7.6 x with per-tensor activation scales; measured here 7.6 x), preference agreement on the decisive pairs
    >= 0.99 (measured 1.0000), Spearman >= 0.997 (measured 0.9986)."""
    # measured (profiles/r04_d_fp8_parity.txt): rms 0.0600 = 7.6 x the noise, 0 flips on the 478 decisive pairs, 0.9863 on all
    # 512, rho 0.99863
    _rank_case_engineered_mxfp8(cuda, "rankset_c1", 8, max_noise_ratio=10.0, min_agree=0.99, min_rho=0.997)


def test_rank_agreement_engineered_c2_mxfp8(cuda):
    """The same statement at the HEADLINE shape: tests/golden/rankeng_c2 - 256 pairs of 8 frames @448^2,
    N = 2186 tokens per video, 240 of them decisive - scored by the mxfp8 FFN path against the reference's bf16 scores, with the
    path's own stated tolerance (same three bounds as @224^2)."""
    _rank_case_engineered_mxfp8(cuda, "rankset_c2", 4, max_noise_ratio=10.0, min_agree=0.99, min_rho=0.997)


# ---- the preset that keeps north_star's bar (round 6; profiles/r06_a_fp8_ffn_subset_study.txt: all 15 subsets of the four FFN
# Linears on both engineered sets - w1 | w3 alone costs more rank agreement than the other three together)
def test_rank_agreement_engineered_c1_mxfp8_rank999(cuda):
    """model.set_ffn_format("mxfp8-rank999") = fc1 + fc2 + w2 on MXFP8 operands (13.7 of the 17.1 TFLOP of FFN per pair; w1 | w3 stays
    bf16), against the REFERENCE's bf16 scores on the engineered rank set @224^2: held to north_star's own bar - Spearman >= 0.999
    over all 1024 scores (measured 0.99920) and NO flip on the 478 decisive pairs - plus rms <= 8 x the reference's bf16 noise
    (measured 6.2 x)."""
    _rank_case_engineered_mxfp8(cuda, "rankset_c1", 8, max_noise_ratio=8.0, min_agree=1.0, min_rho=0.999, fmt="mxfp8-rank999")


def test_rank_agreement_engineered_c2_mxfp8_rank999(cuda):
    """the same preset at the HEADLINE shape (256 pairs @448^2, 240 decisive): Spearman >= 0.999 (measured 0.99912), no flip"""
    _rank_case_engineered_mxfp8(cuda, "rankset_c2", 4, max_noise_ratio=8.0, min_agree=1.0, min_rho=0.999, fmt="mxfp8-rank999")


def test_ffn_fp8_subsets_switch_formats_at_the_seam(cuda):
    """set_ffn_format("mxfp8:<a>+<b>") / presets: a Linear outside the subset runs the bf16 kernels and the seam between an fp8 and a
    bf16 Linear quantises (or not) the very values the all-fp8 path hands over - so (1) the explicit full subset IS the "mxfp8"
    path bit for bit, (2) a one-Linear subset differs from both bf16 and the full set, (3) bad names are refused, (4) bf16 after
    any preset is the bf16 path again."""
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    from mj_video_amd.modeling import FP8_PRESETS
    from util import build_hip_model, make_cfg
    import copy
    from mj_video_amd import configuration as C
    cd = C.tiny_config_dict(56)
    cd["vision_config"].update(hidden_size=128, intermediate_size=512)
    cfg = C.InternVLChatRewardModelingConfig(**cd, **C.mjvideo_head_kwargs())
    model = build_hip_model(cfg, synth.synth_state_dict(cfg, seed=6), cuda)
    px = synth.synth_pixel_values(9, 0, 4, 56).to(cuda)
    ids = synth.synth_input_ids(num_image_tokens_per_tile(cfg) * 4, 1).to(cuda)
    mask = torch.ones_like(ids)

    def hid(fmt, **kw):
        return model.set_ffn_format(fmt, **kw).forward(px, ids, mask).hidden_state.clone()

    bf, full = hid("bf16"), hid("mxfp8")
    assert torch.equal(full, hid("mxfp8:fc1+fc2+w13+w2")) and torch.equal(full, hid("mxfp8", linears=("w2", "w13", "fc2", "fc1")))
    seen = [bf, full]
    for sub in ("fc1", "fc2", "w13", "w2", "fc1+w2", "fc2+w13"):
        h = hid("mxfp8:" + sub)
        assert torch.isfinite(h.float()).all() and all(not torch.equal(h, o) for o in seen), sub
        seen.append(h)
    assert torch.equal(hid("mxfp8-rank999"), hid("mxfp8:fc1+fc2+w2")) and FP8_PRESETS["mxfp8-rank999"] == frozenset(("fc1", "fc2", "w2"))
    assert torch.equal(hid("mxfp8-vit"), hid("mxfp8:fc1+fc2"))
    for bad in ("mxfp8:wq", "mxfp8:"):
        with pytest.raises(ValueError):
            model.set_ffn_format(bad)
    with pytest.raises(ValueError):
        model.set_ffn_format("mxfp8-rank999", linears=("fc1",))
    assert torch.equal(bf, hid("bf16"))
