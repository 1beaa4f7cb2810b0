"""BASELINE configs[4]'s backbone on the MI355X: InternViT + a Phi-3-mini language tower behind the same
InternVLChatRewardModeling API.  Kernels new to it (ABI 7: head_dim 96 attention, rotary embedding in place on a
[q | k | v] projection) against torch references, then the model against fixtures that the REFERENCE's own reward-model
code produced around transformers' Phi3ForCausalLM (tests/golden/make_golden_phi3.py; oracle/ref_phi3.py is pinned to it bit
for bit).  Tolerances: the same noise-floor rules as tests/test_e2e_gpu.py."""
import copy

import numpy as np
import pytest
import torch

from test_e2e_gpu import ATOL_FLOOR, TOL_FACTOR, bits_to_f32, noise_floor, rel_l2
from test_kernels_gpu import assert_close_bf16, attn_reference, rnd
from util import FIELDS, apply_test_overrides, load_golden

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def phi3_cfg(kind, image_size, **llm_over):
    from mj_video_amd import configuration as C
    cd = C.tiny_phi3_config_dict(image_size) if kind == "tiny" else C.internvl2_4b_config_dict(image_size)
    cd["llm_config"].update(llm_over)
    return C.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **C.mjvideo_head_kwargs())


def build_phi3_model(cfg, sd, device):
    from mj_video_amd import synth
    from mj_video_amd.modeling import InternVLChatRewardModeling
    model = InternVLChatRewardModeling.from_config(cfg, dtype=BF)
    model.load_state_dict(sd, strict=True)
    model.config.pad_token_id = synth.PHI3_TOKENS.pad
    model = model.to(BF).to(device)
    model.model.img_context_token_id = synth.PHI3_TOKENS.img_context
    return apply_test_overrides(model.eval())


def phi3_inputs(cfg, videos, pixel_seed, image_size):
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    tk = synth.PHI3_TOKENS
    px, ids = [], []
    for v in videos:
        px.append(synth.synth_pixel_values(pixel_seed, v["video_idx"], v["n_tiles"], image_size))
        ids.append(synth.synth_input_ids(num_image_tokens_per_tile(cfg) * v["n_tiles"], v["caption_seed"],
                                         interleave_frames=v.get("interleave"), tokens=tk))
    ids_b, mask = synth.pad_batch(ids, pad_id=tk.pad)
    return torch.cat(px), ids_b, mask


# ------------------------------------------------------------------------------------------------ kernels
def test_rope_heads_in_place(cuda):
    """mjv_rope_heads_bf16: apply_rotary_pos_emb of transformers' Phi-3 on the q and k heads of a [q | k | v] projection, in place;
    elementwise bf16 arithmetic with bf16 tables: bit-exact against the same torch ops; the v columns are untouched."""
    from mj_video_amd import ops
    rows, H, KV, D = 75, 4, 2, 96
    qkv = rnd(rows, (H + 2 * KV) * D, seed=3)
    pos = (torch.arange(rows, dtype=torch.int32) * 7) % 60
    inv = 1.0 / (1e4 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.arange(64).float()[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), -1)
    cos, sin = (emb.cos() * 1.19).to(BF), (emb.sin() * 1.19).to(BF)
    x = qkv.clone().to(cuda)
    ops.rope_heads(x, H + KV, D, cos.to(cuda), sin.to(cuda), pos.to(cuda))
    t = qkv[:, :(H + KV) * D].view(rows, H + KV, D)

    def rot(a):
        return torch.cat((-a[..., D // 2:], a[..., :D // 2]), -1)

    c, s = cos[pos.long()][:, None, :], sin[pos.long()][:, None, :]
    want = (t * c) + (rot(t) * s)
    assert torch.equal(x[:, :(H + KV) * D].cpu(), want.reshape(rows, -1))
    assert torch.equal(x[:, (H + KV) * D:].cpu(), qkv[:, (H + KV) * D:])
    # a partial rotation (rot_dim < head stride) leaves the rest of the head alone
    y = qkv.clone().to(cuda)
    ops.rope_heads(y, H + KV, D, cos[:, :32].contiguous().to(cuda), sin[:, :32].contiguous().to(cuda), pos.to(cuda))
    c2, s2 = cos[pos.long()][:, None, :32], sin[pos.long()][:, None, :32]
    t2 = t[..., :32]
    want2 = (t2 * c2) + (torch.cat((-t2[..., 16:], t2[..., :16]), -1) * s2)
    got2 = y[:, :(H + KV) * D].cpu().view(rows, H + KV, D)
    assert torch.equal(got2[..., :32], want2) and torch.equal(got2[..., 32:], t[..., 32:])


@pytest.mark.parametrize("kernel", [0, 7])
@pytest.mark.parametrize("D,H,G,causal,lens", [
    (96, 2, 1, True, [150]),
    (96, 4, 1, True, [650, 131, 64, 1]),
    (96, 32, 1, True, [2186]),
    (96, 4, 2, True, [700, 65]),
    (96, 2, 1, False, [17, 17, 17]),
    (96, 4, 1, False, [1025, 577]),
    (96, 2, 1, False, [257, 64, 129]),
])
def test_attention_head_dim_96(cuda, kernel, D, H, G, causal, lens):
    """head_dim 96 (ABI 7; Phi-3-mini's heads) on the round-3 kernel: both of the reference's numerics (mode 1 = eager:
    bf16(bf16(q k^T) * d^-0.5) as modeling_phi3.py:eager_attention_forward rounds; mode 2 = fp32 scores) against the fp32
    reference with the same score rounding, bounds of test_kernels_gpu.test_attention; the two wave counts agree bit for bit."""
    from mj_video_amd import ops
    N = sum(lens)
    KVH = H // G
    q, k, v = rnd(N, H * D, seed=1), rnd(N, KVH * D, seed=2), rnd(N, KVH * D, seed=3)
    scale = float(np.float32(D ** -0.5))
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    for mode in ((1, 2) if causal else (0, 2)):
        out = torch.full((N, H * D), float("nan"), dtype=BF, device=cuda)
        ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu.to(cuda), max(lens), H, G, D, causal, scale, mode, kernel=kernel)
        ref = attn_reference(q, k, v, lens, H, G, D, causal, scale, mode)
        o = out.float().cpu()
        assert torch.isfinite(o).all()
        rel = ((o - ref.float()).norm() / ref.float().norm()).item()
        assert rel < 4e-3, f"mode {mode}: relative L2 error {rel:.3e}"
        assert_close_bf16(out, ref, 2, atol=0.02, what=f"attention d96 mode {mode}")


def test_attention_head_dim_96_in_a_qkv_projection(cuda):
    """the layout the Phi-3 tower launches: q / k / v are column ranges of ONE [rows, (H + 2 KV) * 96] buffer (row stride 9216 at
    4B dims, heads 96 elements apart), O its own buffer - same bits as the launch on separate contiguous tensors."""
    from mj_video_amd import ops
    H, D, lens = 8, 96, [700, 333]
    N = sum(lens)
    qkv = rnd(N, 3 * H * D, seed=11).to(cuda)
    cu = torch.tensor([0, lens[0], N], dtype=torch.int32, device=cuda)
    scale = float(np.float32(D ** -0.5))
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    for mode in (1, 2):
        a = torch.empty(N, H * D, dtype=BF, device=cuda)
        b = torch.empty_like(a)
        ops.attention(q, k, v, a, cu, max(lens), H, 1, D, True, scale, mode)
        ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), b, cu, max(lens), H, 1, D, True, scale, mode)
        assert torch.equal(a, b)


def test_attention_head_dim_96_refuses_the_older_kernels(cuda):
    from mj_video_amd import ops, _lib
    q = torch.zeros(64, 96, dtype=BF, device=cuda)
    cu = torch.tensor([0, 64], dtype=torch.int32, device=cuda)
    for kernel, msg in ((4, "round-3 kernel only"), (5, "round-3 kernel only"), (6, "kernel 6")):
        with pytest.raises(_lib.MjvLibraryError, match=msg):
            ops.attention(q, q, q, torch.empty_like(q), cu, 64, 1, 1, 96, True, 96 ** -0.5, 1, kernel=kernel)
    with pytest.raises(_lib.MjvLibraryError, match="head_dim 80"):
        ops.attention(q[:, :80], q[:, :80], q[:, :80], torch.empty(64, 80, dtype=BF, device=cuda), cu, 64, 1, 1, 80, True, 0.1, 1)


def test_attention_head_dim_96_race_screen(cuda):
    """repeat launches of one problem must agree bit for bit (LDS-DMA double buffer at the new 24 KiB-in-32 KiB tile layout)"""
    from mj_video_amd import ops
    H, D, lens = 32, 96, [2186, 2186, 700]
    N = sum(lens)
    q, k, v = rnd(N, H * D, seed=21).to(cuda), rnd(N, H * D, seed=22).to(cuda), rnd(N, H * D, seed=23).to(cuda)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=cuda)
    first = torch.empty(N, H * D, dtype=BF, device=cuda)
    ops.attention(q, k, v, first, cu, max(lens), H, 1, D, True, 96 ** -0.5, 2)
    for _ in range(20):
        again = torch.empty_like(first)
        ops.attention(q, k, v, again, cu, max(lens), H, 1, D, True, 96 ** -0.5, 2)
        assert torch.equal(first, again)


def test_attention_head_dim_96_long_context(cuda):
    """BASELINE configs[3]'s sequence length on the Phi-3 heads: one causal sequence of 28 810 tokens (16 frames x 7 tiles x 256 +
    text), head_dim 96, both numerics; the fp32 reference is evaluated in query chunks for the first / middle / last rows."""
    from mj_video_amd import ops
    D, H, L = 96, 2, 28810
    q, k, v = rnd(L, H * D, seed=1), rnd(L, H * D, seed=2), rnd(L, H * D, seed=3)
    cu = torch.tensor([0, L], dtype=torch.int32, device=cuda)
    scale = float(np.float32(D ** -0.5))
    rows = torch.cat([torch.arange(0, 70), torch.arange(14000, 14130), torch.arange(L - 200, L)])
    for mode in (1, 2):
        out = torch.empty(L, H * D, dtype=BF, device=cuda)
        ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), out, cu, L, H, 1, D, True, scale, mode)
        o = out.float().cpu()
        assert torch.isfinite(o).all()
        for h in range(H):
            sc = q[rows, h * D:(h + 1) * D].float() @ k[:, h * D:(h + 1) * D].float().t()
            sc = (sc.to(BF).float() * scale).to(BF).float() if mode == 1 else sc * scale
            sc = sc.masked_fill(torch.arange(L)[None, :] > rows[:, None], float("-inf"))
            ref = (torch.softmax(sc, -1).to(BF).float() @ v[:, h * D:(h + 1) * D].float()).to(BF).float()
            got = o[rows, h * D:(h + 1) * D]
            rel = (got - ref).norm() / ref.norm()
            assert rel.item() < 6e-3, (mode, h, rel.item())
            assert (got - ref).abs().max().item() < 0.03


# ------------------------------------------------------------------------------------------------ model
@pytest.mark.parametrize("scores", ["flash", "eager"])
def test_phi3_tiny_cases_against_golden(cuda, scores):
    """tests/golden/phi3_tiny.npz: the reference's reward model around transformers' Phi3ForCausalLM at tiny dims - a single
    video, a sequence SHORTER than the LongRoPE window (short factors, bf16-rounded inv_freq buffer), a right-padded batch, the
    interleaved prompt, a config without rope scaling - every per-layer probe and every output field."""
    from mj_video_amd import synth
    npz, meta = load_golden("phi3_tiny")
    names = [c["name"] for c in meta["cases"]]
    report = []
    for case in meta["cases"]:
        name = case["name"]
        cfg = phi3_cfg("tiny", case["image_size"], **case["llm_overrides"])
        sd = synth.synth_state_dict(cfg, seed=case["weight_seed"], dtype=torch.float32)
        model = build_phi3_model(cfg, sd, cuda)
        model.attention_scores = scores
        px, ids, mask = phi3_inputs(cfg, case["videos"], case["pixel_seed"], case["image_size"])
        assert ids.shape[1] == case["n_tokens"]
        model.debug_probes = {}
        out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
        torch.cuda.synchronize()
        for key, t in model.debug_probes.items():
            if not isinstance(t, torch.Tensor):
                continue
            ref = npz[f"{name}/probe/{key}"]
            got = t.float().cpu().numpy()
            if key.startswith("llm_"):
                lens = [int(m.sum()) for m in mask]
                ref = np.concatenate([ref[b, :lens[b]] for b in range(len(lens))], axis=0)
            assert got.shape == ref.shape, (name, key, got.shape, ref.shape)
            err = rel_l2(got, ref)
            assert np.isfinite(got).all(), (name, key)
            assert err < 0.03, f"{name}:{key} relative L2 error {err:.4f}"
            report.append((name, key, round(err, 5)))
        for f in FIELDS:
            got = getattr(out, f).float().cpu().numpy()
            ref = npz[f"{name}/{f}"]
            assert got.shape == ref.shape, (name, f, got.shape, ref.shape)
            if f in ("hidden_state", "prompt_embedding"):
                assert rel_l2(got, ref) < 0.03, (name, f, rel_l2(got, ref))
                continue
            tol = TOL_FACTOR * noise_floor(npz, names, f) + ATOL_FLOOR
            d = float(np.abs(got - ref).max())
            assert d <= tol, f"{name}:{f} max|d|={d:.4e} > tol {tol:.4e}"
        assert out.score.dtype == torch.float32 and out.rewards.dtype == BF
    print("worst probe errors:", sorted(report, key=lambda r: -r[2])[:5])


@pytest.mark.parametrize("kv_heads,mask_mode", [(1, None), (2, "left"), (4, "holes")])
def test_phi3_gqa_and_masks_against_the_oracle(cuda, kv_heads, mask_mode):
    """what the fixtures do not hold, against the ORACLE (pinned to transformers for the MHA cases) on the same seeded inputs at
    tiny dims: grouped-query heads (4 query heads over 1 / 2 / 4 key-value heads: modeling_phi3.py repeat_kv) and the masks other
    than right padding (left padding, holes: shared host code with the InternLM2 tower, whose reference-executed cases pin it).
    Bounds: the tiny cases' (3 x the fixture's noise floor per field, 3 % on the hidden rows)."""
    from mj_video_amd import synth
    from oracle import ref_phi3
    npz, meta = load_golden("phi3_tiny")
    names = [c["name"] for c in meta["cases"]]
    cfg = phi3_cfg("tiny", 56, hidden_size=384, num_attention_heads=4, num_key_value_heads=kv_heads)
    sd = synth.synth_state_dict(cfg, seed=40 + kv_heads)
    model = build_phi3_model(cfg, sd, cuda)
    model.attention_scores = "eager"          # the oracle's numerics
    vids = [dict(video_idx=0, n_tiles=4, caption_seed=1), dict(video_idx=1, n_tiles=2, caption_seed=2)]
    px, ids, mask = phi3_inputs(cfg, vids, 300, 56)
    ids, mask = synth.remask(ids, mask, mask_mode, pad_id=synth.PHI3_TOKENS.pad)
    out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    ref = ref_phi3.reward_forward(sd, cfg, px, ids, mask, synth.PHI3_TOKENS.img_context, synth.PHI3_TOKENS.pad, cfg.gating_token_pattern)
    for f in FIELDS:
        got, want = getattr(out, f).float().cpu().numpy(), ref[f].float().numpy()
        assert got.shape == want.shape, (f, got.shape, want.shape)
        if f in ("hidden_state", "prompt_embedding"):
            assert rel_l2(got, want) < 0.03, (f, rel_l2(got, want))
            continue
        tol = TOL_FACTOR * noise_floor(npz, names, f) + ATOL_FLOOR
        assert float(np.abs(got - want).max()) <= tol, (f, float(np.abs(got - want).max()), tol)


def test_phi3_trimming_and_prefix_cache_are_invisible(cuda):
    """model.trim_last_layer / model.prefix_cache on the Phi-3 tower (its own projection layout: row slices of qkv_proj for the
    trimmed last layer, whole [q | k | v] prefix rows per layer): with no GEMM slicing K, every CustomOutput field is BIT-IDENTICAL
    with the two switches on (cold and warm cache) and off; the cache serves the second forward; another padded width that crosses
    the LongRoPE window rebuilds it."""
    from mj_video_amd import synth
    cfg = phi3_cfg("tiny", 56, hidden_size=384, num_attention_heads=4, num_key_value_heads=4, original_max_position_embeddings=160)
    model = build_phi3_model(cfg, synth.synth_state_dict(cfg, seed=51), cuda)
    model.use_gemm_workspace = False
    vids = [dict(video_idx=i, n_tiles=t, caption_seed=i) for i, t in enumerate([4, 2, 3])]
    px, ids, mask = (t.to(cuda) for t in phi3_inputs(cfg, vids, 300, 56))

    def run(cache, trim):
        model.prefix_cache, model.trim_last_layer = cache, trim
        return model.forward(px, ids, mask)

    def same(a, b):
        return [f for f in FIELDS if not torch.equal(getattr(a, f), getattr(b, f))]

    plain = run(False, False)
    assert same(plain, run(False, True)) == []                       # queries only where the heads read
    model._prefix = None
    hits = model.prefix_cache_hits
    cold = run(True, True)
    assert model._prefix is not None and model._prefix["P"] == 64 and model.prefix_cache_hits == hits
    warm = run(True, True)
    assert model.prefix_cache_hits == hits + 1
    assert same(plain, cold) == [] and same(plain, warm) == [] and same(plain, run(True, False)) == []
    # a batch whose padded width crosses the original window (160) rotates with the LONG factors: another cache entry
    vids2 = [dict(video_idx=7, n_tiles=8, caption_seed=9)]
    px2, ids2, mask2 = (t.to(cuda) for t in phi3_inputs(cfg, vids2, 300, 56))
    assert ids.shape[1] <= 160 < ids2.shape[1]
    before = model._prefix
    long_on = model.forward(px2, ids2, mask2)
    assert model._prefix is not before
    model.prefix_cache = model.trim_last_layer = False
    assert same(long_on, model.forward(px2, ids2, mask2)) == []


def test_phi3_errors_and_pattern(cuda):
    """the gating rows are found by config.gating_token_pattern (the reference hard-codes the InternLM2 ids, moe_reward.py:45-48):
    a prompt that ends with the InternLM2 pattern raises the reference's ValueError under the Phi-3 config"""
    from mj_video_amd import synth
    cfg = phi3_cfg("tiny", 56)
    model = build_phi3_model(cfg, synth.synth_state_dict(cfg, seed=1), cuda)
    px, ids, mask = phi3_inputs(cfg, [dict(video_idx=0, n_tiles=2, caption_seed=1)], 300, 56)
    bad = ids.clone()
    bad[0, -3:] = torch.tensor([1000, 1001, 1002])
    with pytest.raises(ValueError, match="Token pattern not found"):
        model.forward(px.to(cuda), bad.to(cuda), mask.to(cuda))
    model.norm_fusion = True
    with pytest.raises(NotImplementedError, match="norm_fusion"):
        model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))


@pytest.mark.parametrize("scores", ["flash", "eager"])
@pytest.mark.parametrize("name", ["phi3_layer0_n2186", "phi3_layer31_n2186", "phi3_layer0_n4224"])
def test_phi3_single_layer_at_4b_dims(cuda, name, scores):
    """tests/golden/phi3_layers.npz: ONE Phi3DecoderLayer of transformers at InternVL2-4B dims (hidden 3072, 32 heads x 96, ff 8192)
    on [1, 2186, 3072] rows (short LongRoPE factors) and [1, 4224, 3072] rows (N > 4096: the long ones): sampled output rows
    within 2 x the layer's own bf16-vs-fp32 distance of both of transformers' runs."""
    from util import layer_input_rows, layer_tensors
    from mj_video_amd.modeling import InternVLChatRewardModeling
    npz, meta = load_golden("phi3_layers")
    case = next(c for c in meta["cases"] if c["name"] == name)
    cfg = phi3_cfg("4b", meta["image_size"])
    w = layer_tensors(cfg, f"model.language_model.model.layers.{case['layer']}.", meta["weight_seed"])
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 1
    cfg.llm_config.vocab_size = 128
    model = InternVLChatRewardModeling.from_config(cfg, dtype=BF)
    for prm in model.parameters():
        prm.data.zero_()
    model.model.language_model.model.layers[0].load_state_dict(w, strict=True)
    model = apply_test_overrides(model.to(BF).to(cuda).eval())
    model.attention_scores = scores
    x = layer_input_rows(meta["input_seed"], case["input_tag"], tuple(case["shape"]))
    y = model.run_llm_layer(0, x).float().cpu()
    rows = npz[f"{name}/rows"]
    got, ref, f32 = y[:, rows].numpy(), bits_to_f32(npz[f"{name}/out"]), npz[f"{name}/fp32"]
    noise = case["ref_bf16_vs_fp32"]
    d_ref, d_f32 = rel_l2(got, ref), rel_l2(got, f32)
    print(f"{name} [{scores}]: HIP vs transformers bf16 {d_ref:.5f}, vs its fp32 run {d_f32:.5f}; its bf16 vs fp32 {noise:.5f}")
    assert np.isfinite(got).all()
    assert d_ref <= 2.0 * noise and d_f32 <= 2.0 * noise, (name, d_ref, d_f32, noise)


@pytest.mark.parametrize("size", [224, 448])
def test_phi3_full_4b_dims_against_golden(cuda, size):
    """tests/golden/phi3_full_<size>.npz: InternVL2-4B dims end to end (4.1 G parameters), 8 frames per video, all videos of the
    fixture in one forward: layer probes within 2 x the reference's own bf16-vs-fp32 distance at that layer, hidden rows within 1.5 x the
    reference's own bf16-vs-fp32 distance, head outputs within 3 x its noise floor, rms over the videos within 2 x its rms."""
    from mj_video_amd import synth
    try:
        npz, meta = load_golden(f"phi3_full_{size}")
    except FileNotFoundError:
        pytest.skip("fixture not generated")
    cfg = phi3_cfg("4b", size)
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
    sd["model.language_model.lm_head.weight"] = torch.zeros(1, dtype=BF).expand(cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_phi3_model(cfg, sd, cuda)
    vids = meta["videos"]
    px, ids, mask = phi3_inputs(cfg, vids, meta["pixel_seed"], size)
    model.debug_probes = {}
    out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    probes, rp = model.debug_probes, meta["row_probes"]
    pre = [f"v{v['video_idx']}" for v in vids]
    tiles = vids[0]["n_tiles"]
    cu_rows = np.concatenate([[0], np.cumsum([int(m.sum()) for m in mask])])
    # per layer: the fixture holds the same rows from the fp32 run, i.e. the reference's OWN bf16 noise at that depth (vision
    # 0.4 % -> 1.3 %, decoder 1.0 % -> 4.5 % over its 32 layers at these dims); two samples of one noise differ by sqrt(2) x one:
    # bound 2 x, against the bf16 run and against the fp32 run
    rep = []
    for tower, layers in (("vit", rp["vit_layers"]), ("llm", rp["llm_layers"])):
        for L in layers:
            if tower == "vit":
                got = probes[f"vit_layer{L}"][0, :rp["vit_rows"], :].float().cpu().numpy()
            else:
                got = probes[f"llm_layer{L}"][int(cu_rows[1]) - rp["llm_rows"]:int(cu_rows[1]), :].float().cpu().numpy()
            ref, f32 = bits_to_f32(npz[f"v0/probe/{tower}_layer{L}_rows"]), npz[f"v0/probe/fp32/{tower}_layer{L}_rows"]
            noise, e, e32 = rel_l2(ref, f32), rel_l2(got, ref), rel_l2(got, f32)
            rep.append((f"{tower}{L}", round(e, 4), round(e32, 4), round(noise, 4)))
            assert e <= 2.0 * noise and e32 <= 2.0 * noise, (tower, L, e, e32, noise)
    print("layer probes (layer, HIP vs reference bf16, HIP vs its fp32 run, reference bf16 vs fp32):", rep)
    devs = {"score": [], "aspect_scores": [], "rewards": []}
    for i, p in enumerate(pre):
        for f in FIELDS:
            got, ref = getattr(out, f)[i].float().cpu().numpy(), npz[f"{p}/{f}"][0]
            if f in ("hidden_state", "prompt_embedding"):
                tol = 1.5 * max(rel_l2(npz[f"{q}/{f}"], npz[f"{q}/fp32/{f}"]) for q in pre)
                print(f"{p} {f}: rel-L2 {rel_l2(got, ref):.4f} (bound {tol:.4f})")
                assert rel_l2(got, ref) < tol, (p, f, rel_l2(got, ref), tol)
                continue
            tol = TOL_FACTOR * noise_floor(npz, pre, f) + ATOL_FLOOR
            d = float(np.abs(got - ref).max())
            print(f"{p} {f}: max|d|={d:.3e} tol={tol:.3e}")
            assert d <= tol, (p, f, d, tol)
            if f in devs:
                devs[f].append((got - ref).ravel())
    for f, dv in devs.items():
        rms = float(np.sqrt(np.mean(np.concatenate(dv) ** 2)))
        ref_rms = float(np.sqrt(np.mean(np.concatenate([(npz[f"{p}/{f}"] - npz[f"{p}/fp32/{f}"]).ravel() for p in pre]) ** 2)))
        print(f"{f}: rms(hip - ref) {rms:.4f}; reference bf16-vs-fp32 rms {ref_rms:.4f}")
        assert rms <= 2.0 * ref_rms + ATOL_FLOOR, (f, rms, ref_rms)


def test_phi3_fp8_preset_runs(cuda):
    """the MXFP8 FFN path (gate_up_proj as w1 | w3, down_proj as w2) on the Phi-3 tower at tiny dims: every preset stays within the
    fp8-vs-bf16 distance the InternLM2 tiny cases show (a plumbing check; the rank statements are made on the 2B sets)"""
    from mj_video_amd import synth
    cfg = phi3_cfg("tiny", 56, hidden_size=384, num_attention_heads=4, num_key_value_heads=4, intermediate_size=512)
    cfg.vision_config.hidden_size = 128
    sd = synth.synth_state_dict(cfg, seed=9)
    model = build_phi3_model(cfg, sd, cuda)
    px, ids, mask = phi3_inputs(cfg, [dict(video_idx=0, n_tiles=4, caption_seed=1)], 300, 56)
    ref = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    for fmt in ("mxfp8", "mxfp8-rank999", "mxfp8:w13"):
        out = model.set_ffn_format(fmt).forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
        e = rel_l2(out.hidden_state.float().cpu().numpy(), ref.hidden_state.float().cpu().numpy())
        print(fmt, "hidden_state rel-L2 vs bf16:", round(e, 4))
        assert torch.isfinite(out.score).all() and 0 < e < 0.2
    model.set_ffn_format("bf16")
