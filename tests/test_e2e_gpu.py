"""End-to-end parity on the MI355X: the HIP forward against (a) golden vectors produced by the reference
itself (tests/golden/*.npz, see make_golden.py) and (b) the oracle run on the same seeded inputs.

Tolerance (stated, bf16): the reference's own bf16 run deviates from its fp32 run by a measurable noise
floor (stored next to every golden vector as fp32/*).  The HIP path has the same rounding points but a
different fp32 accumulation order, i.e. it is another sample of that same noise (the reference itself moves by
the same amount when only its CPU thread count changes: full_c1 video 0 scores +1.2804 with 6 threads and +1.2280
with 4).  Every output field must therefore satisfy
    |hip - ref_bf16| <= TOL_FACTOR * noise_floor(field) + ATOL_FLOOR
where noise_floor is max|ref_bf16 - ref_fp32| over every fixture of that model size that holds an fp32 run of the reference
(pooled_noise_floor: a 2-video fixture alone says little about the spread - see its docstring for how far ONE re-associated
fp32 sum moves a score); TOL_FACTOR = 3 because the difference of two noise samples is sqrt(2) larger than one and a
maximum over a few fixture cases under-estimates the maximum over all the elements compared.  The statistical statements
(rms deviation, preference agreement on decisive pairs, rank correlation) are made on the rank sets.
"""
import json
import os

import numpy as np
import pytest
import torch

from util import FIELDS, ROOT, build_hip_model, case_inputs, case_inputs_masked, load_golden, make_cfg

pytestmark = pytest.mark.gpu
TOL_FACTOR = 3.0
ATOL_FLOOR = 2e-3
PER_ELEMENT_SIGMAS = 4.5      # per-element bound of the 2B-dims fixtures, in standard deviations of (hip - reference): _full_case
# fields whose own fp32 sample is six numbers take the rms of the field they are a column of (weighted_scores = the last
# aspect's bf16 score, moe_reward.py:273,294)
_RMS_PROXY = {"weighted_scores": "aspect_scores"}


def noise_floor(npz, prefixes, field):
    return max(float(np.abs(npz[f"{p}/{field}"] - npz[f"{p}/fp32/{field}"]).max()) for p in prefixes
               if f"{p}/fp32/{field}" in npz.files)


_PACKED34 = {"score": slice(0, 1), "aspect_scores": slice(1, 6), "rewards": slice(6, 34)}


def pooled_noise_floor(field):
    """max |ref_bf16 - ref_fp32| of ``field`` over EVERY MJ-VIDEO-2B-dims fixture that holds an fp32 run of the reference
    (full_c1: 4 videos, full_c2: 2, rank sets: 128 + 12 videos for score / aspect_scores / rewards).  A two-video fixture
    under-estimates the reference's own spread badly: the path is chaotic at bf16 - re-associating the fp32 sums of 16 of
    the 16 400 ViT rows of full_c2 (split-K 8 / 4 / 3 / 2 / off in the fc2 tail) moves video 1's score through
    0.43 / 1.02 / 0.72 / 0.86 / 0.65 while the reference's bf16 run sits at 0.73 and its fp32 run at 0.63
    (tools/c2_probe.py) - so the per-video bound uses the largest sample of that spread the fixtures offer."""
    worst = 0.0
    for tag in ("full_c1", "full_c2"):
        try:
            npz, meta = load_golden(tag)
        except FileNotFoundError:
            continue
        pre = [f"v{v['video_idx']}" for v in meta["videos"] if f"v{v['video_idx']}/fp32/{field}" in npz.files]
        if pre:
            worst = max(worst, noise_floor(npz, pre, field))
    if field in _PACKED34:
        for tag in ("rankset_c1", "rankset_c2"):
            try:
                npz, _ = load_golden(tag)
            except FileNotFoundError:
                continue
            b, f = npz["ref_bf16"][..., _PACKED34[field]], npz["ref_fp32"][..., _PACKED34[field]]
            have = ~np.isnan(f[:, 0, 0])
            worst = max(worst, float(np.abs(b[have] - f[have]).max()))
    return worst


def pooled_noise_rms(field):
    """rms of (ref_bf16 - ref_fp32) of ``field`` over every MJ-VIDEO-2B-dims fixture with an fp32 run (same pool as
    pooled_noise_floor)."""
    sq, n = 0.0, 0
    for tag in ("full_c1", "full_c2"):
        npz, meta = load_golden(tag)
        for v in meta["videos"]:
            p = f"v{v['video_idx']}"
            if f"{p}/fp32/{field}" in npz.files:
                dlt = (npz[f"{p}/{field}"] - npz[f"{p}/fp32/{field}"]).astype(np.float64).ravel()
                sq, n = sq + float((dlt ** 2).sum()), n + dlt.size
    for tag in ("rankset_c1", "rankset_c2") if field in _PACKED34 else ():
        try:
            npz, _ = load_golden(tag)
        except FileNotFoundError:
            continue
        b, f = npz["ref_bf16"][..., _PACKED34[field]], npz["ref_fp32"][..., _PACKED34[field]]
        have = ~np.isnan(f[:, 0, 0])
        dlt = (b[have] - f[have]).astype(np.float64).ravel()
        sq, n = sq + float((dlt ** 2).sum()), n + dlt.size
    return float(np.sqrt(sq / max(n, 1)))


def bits_to_f32(a):
    """bf16 bit patterns (uint16) -> float32 values"""
    return (a.astype(np.uint32) << 16).view(np.float32)


# relative-L2 bound of a layer's hidden state against the reference's bf16 run at MJ-VIDEO-2B dims.  The reference's own
# bf16 run is 2.4-2.7 % away from its fp32 run at the END of the language tower (tools/parity_report.py), the error grows
# layer by layer, and two samples of that noise differ by sqrt(2) x one: 3 % for the vision tower (24 layers, LayerScale-
# damped residual updates), 4 % for the language tower's deepest layers.
LAYER_TOL = {"vit": 0.03, "llm": 0.04}
# Per layer (round 4): tests/golden/layers.npz holds the reference's OWN one-layer noise at production shape -
# its bf16 run against its fp32 run: 0.30 % per vision layer, 0.44 % per language layer - and the whole-row probes of the full
# fixtures show it compounding like a random walk (HIP vs reference, measured: vision 0.19 / 0.70 / 0.98 / 1.18 / 1.36 % at layers
# 0 / 5 / 11 / 17 / 23 = 0.30 % x sqrt(L + 1) within 10 %; language 0.65 / 1.65 / 2.3 / 2.8 % at 0 / 7 / 15 / 23, which enters
# with the 0.48 % the image rows bring along).  Bound = 2 x that model (two samples of one noise differ by sqrt(2) x one, and
# attention mixing runs the language tower ~20 % above the pure random walk), never above the flat figure.
ONE_LAYER_NOISE = {"vit": 0.0030, "llm": 0.0044}
LLM_ENTRY_NOISE = 0.0048


def layer_tol(tower, L):
    base = LLM_ENTRY_NOISE if tower == "llm" else 0.0
    return min(2.0 * float(np.sqrt(ONE_LAYER_NOISE[tower] ** 2 * (L + 1) + base ** 2)), LAYER_TOL[tower])


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("scores,fused", [("flash", False), ("flash", True), ("eager", False)])
def test_tiny_cases_against_golden(cuda, scores, fused):
    """(the shipped default - flash scores, norms not folded - first; then both of the reference's attention numerics -
    model.attention_scores - and the folded norms against the same reference-executed fixtures)"""
    from mj_video_amd import synth
    npz, meta = load_golden("tiny")
    names = [c["name"] for c in meta["cases"]]
    report = []
    for case in meta["cases"]:
        name = case["name"]
        cfg = make_cfg("tiny", case["image_size"], case["vit_image_size"])
        sd = synth.synth_state_dict(cfg, seed=case["weight_seed"], dtype=torch.float32)
        model = build_hip_model(cfg, sd, cuda)
        model.attention_scores = scores
        model.norm_fusion = fused
        px, ids, mask, _ = case_inputs_masked(cfg, case)   # (round 6: "leftpad" / "holes" = masks other than right padding)
        model.debug_probes = {}
        out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
        torch.cuda.synchronize()
        # layer-by-layer first: localises a broken kernel instead of a vague score mismatch
        for key, t in model.debug_probes.items():
            if not isinstance(t, torch.Tensor):
                continue   # operand captures (llm_attn0) are for the attention cross-checks, not layer states
            ref = npz[f"{name}/probe/{key}"]
            got = t.float().cpu().numpy()
            if key.startswith("llm_"):
                # golden is padded [B, N, C]; ours is packed: the rows at the valid columns, in order
                ref = np.concatenate([ref[b][mask[b].bool().numpy()] for b in range(mask.shape[0])], axis=0)
            assert got.shape == ref.shape, (name, key, got.shape, ref.shape)
            err = rel_l2(got, ref)
            assert np.isfinite(got).all(), (name, key)
            assert err < 0.03, f"{name}:{key} relative L2 error {err:.4f}"
            report.append((name, key, err))
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
        assert out.score.dtype == torch.float32 and out.aspect_scores.dtype == torch.float32
        assert out.rewards.dtype == torch.bfloat16
    worst = sorted(report, key=lambda r: -r[2])[:5]
    print("worst probe errors:", worst)


def _full_case(cuda, tag, image_size):
    from mj_video_amd import synth
    npz, meta = load_golden(tag)
    cfg = make_cfg("2b", image_size)
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)  # loaded (strict) but never read
    model = build_hip_model(cfg, sd, cuda)
    with_fp32 = [f"v{v['video_idx']}" for v in meta["videos"] if f"v{v['video_idx']}/fp32/score" in npz.files]
    # batch all videos of the fixture in ONE forward (the reference ran them one by one: packing must not matter)
    px, ids, mask, _ = case_inputs(cfg, meta["videos"], meta["pixel_seed"], image_size)
    model.debug_probes = {}
    out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    probes = model.debug_probes
    tiles = meta["videos"][0]["n_tiles"]
    cu_rows = np.concatenate([[0], np.cumsum([int(m.sum()) for m in mask])])
    layer_report, dev_by_field = [], {f: [] for f in _PACKED34}
    # hidden-state rows the heads read: the reference's bf16 run is itself 2.2-2.6 % (relative L2) away from its fp32 run at
    # these rows; two independent samples of that noise differ by sqrt(2) x as much, so the HIP path is held to 1.5 x the
    # largest bf16-vs-fp32 distance the fixture recorded for the field (a re-associated fp32 sum - e.g. a K-sliced GEMM -
    # moves these rows by about 1 % of their norm without moving them away from the fp32 truth)
    hid_tol = {f: 1.5 * max(rel_l2(npz[f"{q}/{f}"], npz[f"{q}/fp32/{f}"]) for q in with_fp32)
               for f in ("hidden_state", "prompt_embedding")}
    for i, v in enumerate(meta["videos"]):
        p = f"v{v['video_idx']}"
        if f"{p}/probe/vit_embed_head" in npz.files:
            sl = slice(i * tiles, (i + 1) * tiles)
            for key, src in (("vit_embed_head", "vit_embed"), ("vit_layer0_head", "vit_layer0"), ("vit_embeds_head", "vit_embeds")):
                got = probes[src][sl, :4, :16].float().cpu().numpy()
                assert rel_l2(got, npz[f"{p}/probe/{key}"]) < 0.03, (tag, p, key, rel_l2(got, npz[f"{p}/probe/{key}"]))
            # language tower: the last 4 token rows of the first and the last decoder layer (packed rows of this video)
            lo_row, hi_row = int(cu_rows[i]), int(cu_rows[i + 1])
            nl = cfg.llm_config.num_hidden_layers
            for key, src in (("llm_layer0_tail", "llm_layer0"), ("llm_last_tail", f"llm_layer{nl - 1}")):
                got = probes[src][hi_row - 4:hi_row, :16].float().cpu().numpy()
                e = rel_l2(got, npz[f"{p}/probe/{key}"])
                assert e < LAYER_TOL[src.split("_")[0]], (tag, p, key, e)
            # whole-row samples of several layers of both towers at MJ-VIDEO-2B dims (thousands of elements per probe)
            rp = meta.get("row_probes")
            if rp and f"{p}/probe/vit_layer{rp['vit_layers'][0]}_rows" in npz.files:
                for L in rp["vit_layers"]:
                    ref = bits_to_f32(npz[f"{p}/probe/vit_layer{L}_rows"])
                    got = probes[f"vit_layer{L}"][i * tiles, :rp["vit_rows"], :].float().cpu().numpy()
                    e = rel_l2(got, ref)
                    layer_report.append((tag, p, f"vit{L}", round(e, 4)))
                    assert e < layer_tol("vit", L), (tag, p, "vit_layer", L, e, layer_tol("vit", L))
                for L in rp["llm_layers"]:
                    ref = bits_to_f32(npz[f"{p}/probe/llm_layer{L}_rows"])
                    got = probes[f"llm_layer{L}"][hi_row - rp["llm_rows"]:hi_row, :].float().cpu().numpy()
                    e = rel_l2(got, ref)
                    layer_report.append((tag, p, f"llm{L}", round(e, 4)))
                    assert e < layer_tol("llm", L), (tag, p, "llm_layer", L, e, layer_tol("llm", L))
        for f in FIELDS:
            got = getattr(out, f)[i].float().cpu().numpy()
            ref = npz[f"{p}/{f}"][0]
            if f in ("hidden_state", "prompt_embedding"):
                print(f"{tag} {p} {f}: rel-L2 {rel_l2(got, ref):.4f} (bound {hid_tol[f]:.4f})")
                assert rel_l2(got, ref) < hid_tol[f], (tag, p, f, rel_l2(got, ref), hid_tol[f])
                continue
            # per-element bound (round 6, VERDICT r5 "weak" 2: 3 x the pooled MAXIMUM of the reference's noise let a kernel five
            # times noisier pass - score tol 0.83 against observed deviations of 0.02 - 0.22): the HIP value and the reference's
            # bf16 value are two samples of one noise around the fp32 truth, so their difference has standard deviation
            # sqrt(2) x the reference's pooled bf16-vs-fp32 rms of the field; bound = PER_ELEMENT_SIGMAS of that (4.5: 7e-6 per
            # element - a few hundred elements per run), never above the old maximum-based bound.  Measured envelope over the 8
            # fixture videos (profiles/r05_u_parity_statements.txt): score 0.22 (bound 0.43), aspect_scores 0.27 (0.40),
            # rewards 0.21 (0.35).
            tol_max = TOL_FACTOR * max(noise_floor(npz, with_fp32, f), pooled_noise_floor(f)) + ATOL_FLOOR
            tol = min(tol_max, PER_ELEMENT_SIGMAS * np.sqrt(2.0) * pooled_noise_rms(_RMS_PROXY.get(f, f)) + ATOL_FLOOR)
            d = float(np.abs(got - ref).max())
            print(f"{tag} {p} {f}: max|d|={d:.3e} tol={tol:.3e} (maximum-based bound of rounds 1-5: {tol_max:.3e})")
            assert d <= tol, f"{tag}:{p}:{f} max|d|={d:.4e} > tol {tol:.4e}"
            if f in dev_by_field:
                dev_by_field[f].append((got - ref).ravel())
    # the statements that bite at MJ-VIDEO-2B dims: over the fixture's videos the HIP path deviates from the reference's
    # bf16 run, in the rms sense, by no more than 2 x what the reference's bf16 run deviates from its own fp32 run
    # (the difference of two samples of one noise is sqrt(2) x one sample), and every probed layer stays within LAYER_TOL
    for f, devs in dev_by_field.items():
        rms = float(np.sqrt(np.mean(np.concatenate(devs) ** 2)))
        ref_rms = pooled_noise_rms(f)
        print(f"{tag} {f}: rms(hip - ref) over {len(devs)} videos = {rms:.4f}; reference bf16-vs-fp32 rms = {ref_rms:.4f}")
        assert rms <= 2.0 * ref_rms + ATOL_FLOOR, (tag, f, rms, ref_rms)
    print("layer probes (relative L2 vs the reference):", layer_report)
    return model, cfg


def test_full_c1_against_golden(cuda):
    """MJ-VIDEO-2B dims, 8 frames @224 (BASELINE.json configs[0] shape), 4 videos batched"""
    _full_case(cuda, "full_c1", 224)


def test_full_c2_against_golden(cuda):
    """MJ-VIDEO-2B dims, 8 frames @448, N = 2186 (BASELINE.json configs[1] shape), 4 videos batched"""
    _full_case(cuda, "full_c2", 448)


_RANK_CACHE = {}
_RANK_INPUTS = {}


def _head_checksum(sd):
    import hashlib
    return hashlib.sha1(b"".join(sd[k].contiguous().view(torch.int16).numpy().tobytes() for k in sorted(sd))).hexdigest()


def _rank_run(cuda, name, pairs_per_forward, ffn_format="bf16"):
    """Scores every pair of rank set ``name`` ONCE per session (the backbone pass is the expensive part) and returns
    dict(meta, got [P,2,34] under the default synthetic heads, eng [P,2,34] under the engineered heads of
    rankeng_* - computed by ``heads_forward`` on the hidden rows of the very same forwards - or None)."""
    key = name if ffn_format == "bf16" else f"{name}:{ffn_format}"
    if key in _RANK_CACHE:
        return _RANK_CACHE[key]
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    try:
        npz, meta = load_golden(name)
    except FileNotFoundError:
        pytest.skip(f"{name} fixture not generated")
    P = npz["ref_bf16"].shape[0]
    cfg = make_cfg("2b", meta["image_size"])
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    model.set_ffn_format(ffn_format)
    nt = meta["n_tiles"]
    H = cfg.llm_config.hidden_size
    got = np.zeros((P, 2, 34), dtype=np.float32)
    hr = torch.empty(P * 2, H, dtype=torch.bfloat16, device=cuda)
    hg = torch.empty(P * 2, H, dtype=torch.bfloat16, device=cuda)
    PB = pairs_per_forward
    # the synthetic inputs of a set are generated on the host (seconds per hundred videos) and kept for the set's second
    # scoring (the mxfp8 run of tests/test_fp8_gpu.py scores the same videos): 2.5 GB @224^2, 4.9 GB @448^2 of host memory
    batches = _RANK_INPUTS.setdefault((name, PB), [])
    for bi, p0 in enumerate(range(0, P, PB)):
        if bi == len(batches):
            px, ids = [], []
            for p in range(p0, min(P, p0 + PB)):
                row = synth.synth_input_ids(num_image_tokens_per_tile(cfg) * nt, caption_seed=meta["caption_seed_base"] + p)
                for j in range(2):
                    px.append(synth.synth_pixel_values(meta["pixel_seed"], 2 * p + j, nt, meta["image_size"]))
                    ids.append(row)
            ids_b, mask = synth.pad_batch(ids)
            batches.append((torch.cat(px), ids_b, mask, len(px)))
        px_b, ids_b, mask, n = batches[bi]
        out = model.forward(px_b.to(cuda), ids_b.to(cuda), mask.to(cuda))
        got[p0:p0 + n // 2] = model.last_packed34.float().cpu().numpy().reshape(-1, 2, 34)
        hr[2 * p0:2 * p0 + n] = out.hidden_state
        hg[2 * p0:2 * p0 + n] = out.prompt_embedding
    eng = None
    eng_name = name.replace("rankset", "rankeng")
    try:
        enpz, emeta = load_golden(eng_name)
    except FileNotFoundError:
        enpz = None
    if enpz is not None:
        w = torch.from_numpy(bits_to_f32(enpz["regression_weight_bits"])).to(torch.bfloat16)
        head_sd = synth.engineered_head_state_dict(cfg, emeta["weight_seed"], w, enpz["gate_dirs"])
        # the engineered heads are rebuilt here from the stored factors: they must be the very weights the reference scored
        assert _head_checksum(head_sd) == emeta["head_weights_sha1"], "engineered head weights differ from the fixture's"
        res = model.load_state_dict(head_sd, strict=False)
        assert not res.unexpected_keys
        eng = np.zeros((P, 2, 34), dtype=np.float32)
        for r0 in range(0, 2 * P, 256):
            model.heads_forward(hr[r0:r0 + 256], hg[r0:r0 + 256])
            blk = model.last_packed34.float().cpu().numpy()
            eng.reshape(-1, 34)[r0:r0 + blk.shape[0]] = blk
    del model
    torch.cuda.empty_cache()
    _RANK_CACHE[key] = dict(meta=meta, got=got, eng=eng)
    return _RANK_CACHE[key]


def test_rank_agreement_c1(cuda):
    """512 pairs at MJ-VIDEO-2B dims, 8 frames @224 (BASELINE.json configs[0] shape), default synthetic heads: the
    noise-model test (near-ties included)"""
    _rank_case(cuda, "rankset_c1", pairs_per_forward=8)


def test_rank_agreement_c2(cuda):
    """Pairs at the headline shape (8 frames @448, N = 2186, BASELINE.json configs[1]), default synthetic heads; the set is
    smaller because every pair costs the reference about a minute of CPU time."""
    _rank_case(cuda, "rankset_c2", pairs_per_forward=4)


def test_rank_agreement_engineered_c1(cuda):
    _rank_case_engineered(cuda, "rankset_c1", pairs_per_forward=8)


def test_rank_agreement_engineered_c2(cuda):
    _rank_case_engineered(cuda, "rankset_c2", pairs_per_forward=4)


def _rank_case_engineered(cuda, name, pairs_per_forward):
    """north_star: >= 0.999 rank agreement with the reference on a fixed synthetic set - on the WHOLE set.
    tests/golden/rankeng_*.npz is the rank set re-scored by the reference's own head code under heads that are fit to the
    backbone the way trained heads are (make_golden.gen_rankset_eng): score differences between videos are two orders of
    magnitude above the reference's bf16 noise, and every stored pair is decisive by construction (margin > 12 x the
    rms bf16-vs-fp32 deviation of these very scores).  Asserted on all kept pairs / all their scores:
      pairwise preference agreement >= 0.999, Spearman rho >= 0.999, good/bad flag agreement >= 0.999 away from zero,
      rms deviation <= 2 x the reference's own, and prefer_Acc / Acc of the eval protocol (eval_genai_mjvideo.py:142-165,
      vote types assigned by pair index) equal to the counts the reference's bookkeeping code produced from its scores."""
    from scipy.stats import spearmanr
    from mj_video_amd import harness
    run = _rank_run(cuda, name, pairs_per_forward)
    if run["eng"] is None:
        pytest.skip(f"{name.replace('rankset', 'rankeng')} fixture not generated")
    enpz, emeta = load_golden(name.replace("rankset", "rankeng"))
    ref, keep, got = enpz["ref_bf16"], enpz["keep"], run["eng"]
    P = ref.shape[0]
    assert got.shape[0] >= P
    got = got[:P]      # (an engineered set may hold fewer pairs than the rank set it was cut from)
    f32, idx32 = enpz["ref_fp32"], enpz["fp32_pairs"]
    noise = (ref[idx32][..., 0] - f32[..., 0]).ravel()
    noise_rms = float(np.sqrt((noise ** 2).mean()))
    d = (got[..., 0] - ref[..., 0]).ravel()
    spread = float(ref[..., 0].std())
    print(f"{name} engineered: pairs kept {int(keep.sum())}/{P}; score spread {spread:.4f}; reference bf16-vs-fp32 rms {noise_rms:.5f}; "
          f"|hip - ref| rms {np.sqrt((d ** 2).mean()):.5f} max {np.abs(d).max():.5f}")
    assert keep.sum() >= 0.85 * P
    assert np.sqrt((d ** 2).mean()) <= 2.0 * noise_rms + 1e-3
    agree = np.sign(got[:, 0, 0] - got[:, 1, 0]) == np.sign(ref[:, 0, 0] - ref[:, 1, 0])
    print(f"preference agreement on the kept pairs {agree[keep].mean():.5f} ({int((~agree[keep]).sum())} flips), on all pairs {agree.mean():.5f}")
    assert agree[keep].mean() >= 0.999
    rho = spearmanr(got[keep][..., 0].ravel(), ref[keep][..., 0].ravel()).correlation
    rho_all = spearmanr(got[..., 0].ravel(), ref[..., 0].ravel()).correlation
    print(f"spearman rho over the kept pairs' scores {rho:.6f}, over all {2 * P} scores {rho_all:.6f}")
    assert rho >= 0.999 and rho_all >= 0.999
    far = np.abs(ref[..., 0]) > 12 * noise_rms
    assert ((got[..., 0] > 0) == (ref[..., 0] > 0))[far].mean() >= 0.999
    # eval protocol on the kept pairs: same counts as the reference's own bookkeeping code gave for its own scores
    cyc = ["leftvote", "rightvote", "tievote", "bothbad_vote"]
    counts = harness.evaluate_votes((cyc[i % 4], float(got[i, 0, 0]), float(got[i, 1, 0])) for i in range(P) if keep[i])
    exp = emeta["accuracy_on_kept_pairs"]
    print(f"prefer_Acc {counts.prefer_acc:.4f} Acc {counts.acc:.4f}; reference counts {exp}")
    # the preference votes (decisive pairs by construction) must give the reference's counts exactly; a tie / both-bad vote
    # is decided by the SIGN of a score, so a kept pair with a reference score within 12 x the noise rms of zero can fall
    # either way under any re-association of the fp32 sums (round 3: the attention kernel's exact-power-of-two softmax
    # offsets moved one such vote): the truth count may differ by at most the number of such borderline votes
    assert (counts.prefer_truth, counts.prefer_total, counts.total) == (exp["prefer_truth"], exp["prefer_total"], exp["total"])
    border = sum(1 for i in range(P) if keep[i] and cyc[i % 4] in ("tievote", "bothbad_vote")
                 and min(abs(float(ref[i, 0, 0])), abs(float(ref[i, 1, 0]))) < 12 * noise_rms)
    print(f"truth {counts.truth} vs reference {exp['truth']}; borderline tie / both-bad votes: {border}")
    assert abs(counts.truth - exp["truth"]) <= border
    exact = harness.evaluate_votes((cyc[i % 4], float(got[i, 0, 0]), float(got[i, 1, 0])) for i in range(P) if keep[i]
                                   and not (cyc[i % 4] in ("tievote", "bothbad_vote")
                                            and min(abs(float(ref[i, 0, 0])), abs(float(ref[i, 1, 0]))) < 12 * noise_rms))
    ref_exact = harness.evaluate_votes((cyc[i % 4], float(ref[i, 0, 0]), float(ref[i, 1, 0])) for i in range(P) if keep[i]
                                       and not (cyc[i % 4] in ("tievote", "bothbad_vote")
                                                and min(abs(float(ref[i, 0, 0])), abs(float(ref[i, 1, 0]))) < 12 * noise_rms))
    assert (exact.prefer_truth, exact.truth, exact.total) == (ref_exact.prefer_truth, ref_exact.truth, ref_exact.total)


def _rank_case(cuda, name, pairs_per_forward):
    """Fixed synthetic set of pairs scored by the reference under the DEFAULT synthetic heads (random rows: the
    reference's own bf16 noise is ~7 % of the score spread, so this set has near-ties; it is the noise-model test, the
    whole-set >= 0.999 criterion is asserted on the engineered set): the HIP scores must (1) deviate from the reference no
    more than the reference deviates from its own fp32 run, (2) give the same pairwise preference on >= 0.999 of the
    decisive pairs (margin > 12 x the reference's noise rms) and on no fewer of ALL pairs than the reference's noise
    predicts, (3) the same good/bad flag, (4) rank mutually separated scores identically; near-ties are reported, not hidden."""
    run = _rank_run(cuda, name, pairs_per_forward)
    npz, meta = load_golden(name)
    ref = npz["ref_bf16"]
    ref32 = npz["ref_fp32"]
    P = ref.shape[0]
    got = run["got"]
    ref = ref[:P]
    f32 = ref32[:P]
    have32 = ~np.isnan(f32[:, 0, 0])
    noise = np.abs(ref[have32][..., 0] - f32[have32][..., 0])
    noise_max, noise_rms = float(noise.max()), float(np.sqrt((noise ** 2).mean()))
    d = np.abs(got[..., 0] - ref[..., 0])
    d32 = np.abs(got[have32][..., 0] - f32[have32][..., 0])
    print(f"pairs={P}  reference noise (bf16 vs fp32 score, {int(have32.sum()) * 2} scores): max={noise_max:.3e} "
          f"rms={noise_rms:.3e}   |hip-ref|: max={d.max():.3e} rms={np.sqrt((d ** 2).mean()):.3e}   "
          f"|hip-fp32|: max={d32.max():.3e} rms={np.sqrt((d32 ** 2).mean()):.3e}")
    # (1) the HIP path is statistically no further from the reference than the reference is from its own fp32 run
    assert np.sqrt((d ** 2).mean()) <= 2.0 * noise_rms + ATOL_FLOOR
    # the worst single video: 3 x the largest reference deviation seen, or 8 x its rms when only a few fp32 runs exist
    # (the maximum of 12 samples says little about the maximum of 96; see pooled_noise_floor about the heavy tail)
    assert d.max() <= max(TOL_FACTOR * noise_max, 8.0 * noise_rms) + ATOL_FLOOR
    # (2) pairwise preference on the decisive pairs of the fixed set: a sign can only flip when the margin is below the
    #     sum of the two videos' errors, so the metric is defined on pairs whose margin is far above the noise: 12 x the
    #     reference's bf16-vs-fp32 rms (about 4 x the largest deviation seen; the rms, unlike the maximum, does not grow
    #     with the number of fp32 runs in the fixture)
    margin = np.abs(ref[:, 0, 0] - ref[:, 1, 0])
    decisive = margin > 12 * noise_rms
    agree = np.sign(got[:, 0, 0] - got[:, 1, 0]) == np.sign(ref[:, 0, 0] - ref[:, 1, 0])
    self_agree = np.sign(ref[have32][:, 0, 0] - ref[have32][:, 1, 0]) == np.sign(f32[have32][:, 0, 0] - f32[have32][:, 1, 0])
    print(f"decisive pairs {int(decisive.sum())}/{P} (margin > {12 * noise_rms:.3f} = {12 * noise_rms / noise_max:.1f} x max noise); "
          f"preference agreement: decisive={agree[decisive].mean():.4f} all={agree.mean():.4f}; "
          f"reference bf16-vs-fp32 self-agreement on {int(have32.sum())} pairs={self_agree.mean():.4f}")
    assert decisive.sum() >= 0.4 * P, "synthetic set has too many near-ties to be meaningful"
    assert agree[decisive].mean() >= 0.999
    # all pairs, near-ties included: a pair flips when the two videos' deviations (each ~ sqrt(2) x the reference's noise,
    # so 2 x noise_rms on the margin) exceed its margin; the agreement must not fall below what that noise model predicts
    from scipy.stats import norm
    p_flip = norm.cdf(-margin / (2.0 * noise_rms))
    expected = 1.0 - p_flip.mean()
    sigma = float(np.sqrt((p_flip * (1 - p_flip)).sum())) / P
    print(f"all-pairs agreement {agree.mean():.4f}; predicted from the reference's own noise {expected:.4f} +- {sigma:.4f}")
    assert agree.mean() >= expected - 3.0 * sigma
    # (3) good/bad flag (score > 0) away from zero, (4) rank correlation over all 2P scores
    good = (got[..., 0] > 0) == (ref[..., 0] > 0)
    far = np.abs(ref[..., 0]) > 6 * noise_rms
    assert good[far].mean() >= 0.999
    from scipy.stats import spearmanr
    rho = spearmanr(got[..., 0].ravel(), ref[..., 0].ravel()).correlation
    rho_self = spearmanr(ref[have32][..., 0].ravel(), f32[have32][..., 0].ravel()).correlation
    print(f"spearman rho(hip, ref)={rho:.5f}   reference bf16-vs-fp32 rho={rho_self:.5f}")
    assert rho >= min(0.99, rho_self - 0.005)
    # well-separated scores (greedy selection with gaps > 12 x noise rms) must be ranked identically
    order = np.argsort(ref[..., 0].ravel())
    keep, last = [], -np.inf
    for i in order:
        if ref[..., 0].ravel()[i] - last > 12 * noise_rms:
            keep.append(i)
            last = ref[..., 0].ravel()[i]
    sep = spearmanr(got[..., 0].ravel()[keep], ref[..., 0].ravel()[keep]).correlation
    print(f"{len(keep)} mutually separated scores: spearman={sep:.5f}")
    assert sep >= 0.999


@pytest.mark.parametrize("fused", [False, True])
def test_last_layer_trimming_is_invisible(cuda, fused):
    """production path (last decoder layer evaluated only on the 2 rows per sample the heads read) vs the full evaluation
    of every row (the debug-probe mode): every output field bit-identical at these shapes when both runs sum K in one
    order (K-slicing of under-filled GEMMs off: the 2 700-row full evaluation would otherwise run its K = 8192 GEMM as two
    slices while the 10 trimmed rows never do); with slicing on, the two differ like any re-associated fp32 sum"""
    from mj_video_amd import synth, ops
    cfg = make_cfg("2b", 224)
    sd = synth.synth_state_dict(cfg, seed=0, lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    model.norm_fusion = fused     # (norms folded into their GEMMs: the trimmed rows take the same folded form)
    vids = [dict(video_idx=i, n_tiles=t, caption_seed=i) for i, t in enumerate([8, 6, 8, 3, 5])]
    px, ids, mask, _ = case_inputs(cfg, vids, 77, 224)
    px, ids, mask = px.to(cuda), ids.to(cuda), mask.to(cuda)
    try:
        model.use_gemm_workspace = False      # no split-K scratch: no GEMM of this forward slices K
        one = model.forward(px, ids, mask)
        model.debug_probes = {}
        full = model.forward(px, ids, mask)
        model.debug_probes = None
        for f in FIELDS:
            assert torch.equal(getattr(one, f), getattr(full, f)), f
    finally:
        model.debug_probes = None
        model.use_gemm_workspace = True
    sliced = model.forward(px, ids, mask)
    for f in ("hidden_state", "prompt_embedding"):
        e = rel_l2(getattr(sliced, f).float().cpu().numpy(), getattr(one, f).float().cpu().numpy())
        print(f"K-sliced vs unsliced {f}: relative L2 {e:.4f}")
        # not small: one re-associated fp32 sum in the last layers' GEMMs moves these rows as far as the reference's own bf16
        # run is from its fp32 run (2.2-2.6 %); two samples of that noise differ by up to sqrt(2) x as much
        assert e < 0.045, (f, e)


@pytest.mark.parametrize("fmt", ["bf16", "mxfp8"])
def test_prefix_cache_is_invisible_and_invalidates(cuda, fmt):
    """``model.prefix_cache``: the keys / values of the constant prompt prefix come from a pass over the
    prefix tokens alone, run by the first forward that meets the prefix and reused by the next ones; every forward leaves
    those rows out of the language tower and attends to the cached rows.  At 2B dims @224^2 (5 videos of different lengths,
    then another batch):
      * with no GEMM slicing K, a forward with the cache (cold or warm) is bit-identical to ``prefix_cache = False`` in every
        field (a row's sums, a query's softmax do not depend on which other rows the launch holds); with K-slicing on the
        two differ like any re-associated fp32 sum - and cold and warm forwards are bit-identical to each other either way;
      * the cached K / V rows are bit-identical to the prefix rows an uncached forward of ANOTHER batch computes (layer 0);
      * invalidation: new weights (load_state_dict), another prefix, another attention numerics setting -> recomputed."""
    from mj_video_amd import synth
    cfg = make_cfg("2b", 224)
    sd = synth.synth_state_dict(cfg, seed=0, lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    model.set_ffn_format(fmt)
    vids = [dict(video_idx=i, n_tiles=t, caption_seed=i) for i, t in enumerate([8, 6, 8, 3, 5])]
    px, ids, mask, _ = case_inputs(cfg, vids, 77, 224)
    px, ids, mask = px.to(cuda), ids.to(cuda), mask.to(cuda)
    vids2 = [dict(video_idx=10 + i, n_tiles=t, caption_seed=20 + i) for i, t in enumerate([2, 8, 7])]
    px2, ids2, mask2, _ = case_inputs(cfg, vids2, 78, 224)
    px2, ids2, mask2 = px2.to(cuda), ids2.to(cuda), mask2.to(cuda)

    def fields_equal(a, b):
        return [f for f in FIELDS if not torch.equal(getattr(a, f), getattr(b, f))]

    try:
        model.use_gemm_workspace = False
        model.prefix_cache = False
        ref, ref2 = model.forward(px, ids, mask), model.forward(px2, ids2, mask2)
        assert model._prefix is None and model.prefix_cache_hits == 0
        model.prefix_cache = True
        first = model.forward(px, ids, mask)
        assert model.prefix_cache_hits == 0 and model._prefix is not None and model._prefix["P"] == 64
        assert fields_equal(first, ref) == []
        cached_k = [t.clone() for t in model._prefix["k"]]
        cached_v_last = model._prefix["v_last"].clone()
        assert len(cached_k) == cfg.llm_config.num_hidden_layers
        again, other = model.forward(px, ids, mask), model.forward(px2, ids2, mask2)
        assert model.prefix_cache_hits == 2
        assert fields_equal(again, ref) == [] and fields_equal(other, ref2) == []
        # the prefix rows of ANOTHER batch, computed inside an uncached full forward, are the cached rows (first decoder layer)
        model.debug_probes = {}
        model.forward(px2, ids2, mask2)
        probe, model.debug_probes = model.debug_probes["llm_attn0"], None
        assert torch.equal(probe["k"][:64], cached_k[0])
        G2 = cfg.llm_config.num_attention_heads // cfg.llm_config.num_key_value_heads + 2
        # (values only: the columns between two value heads of the wqkv output buffer hold un-rotated q / k of the small-tile rows
        # or nothing at all - the 256-tile kernel's rotary epilogue sends q / k straight to their own buffers)
        KVH = cfg.llm_config.num_key_value_heads
        assert torch.equal(probe["v"][:64].reshape(64, -1)[:, : (KVH - 1) * G2 * 128 + 128].unfold(1, 128, G2 * 128),
                           model._prefix["v"][0][:, (G2 - 1) * 128:].unfold(1, 128, G2 * 128))
        assert torch.equal(model._prefix["v_last"].view(64, -1, 2, 128)[:, :, 1], model._prefix["v"][-1].view(64, -1, G2, 128)[:, :, G2 - 1])
        # a rebuilt cache holds the same rows
        model._prefix = None
        model.forward(px2, ids2, mask2)
        assert all(torch.equal(a, b) for a, b in zip(cached_k, model._prefix["k"])) and torch.equal(cached_v_last, model._prefix["v_last"])
        # invalidation 1: another attention numerics setting
        hits = model.prefix_cache_hits
        model.attention_scores = "eager"
        eager = model.forward(px, ids, mask)
        assert model.prefix_cache_hits == hits and model._prefix["settings"][2] == "eager"
        model.prefix_cache = False
        assert fields_equal(eager, model.forward(px, ids, mask)) == []
        model.prefix_cache = True
        model.attention_scores = "flash"
        # invalidation 2: another prefix (one system token changed in every sample)
        model._prefix_misses = 0          # (the thrash guard, tested below, counts consecutive misses)
        model.forward(px, ids, mask)
        hits = model.prefix_cache_hits
        ids3 = ids.clone()
        ids3[:, 7] += 1
        changed = model.forward(px, ids3, mask)
        assert model.prefix_cache_hits == hits
        assert model.forward(px, ids3, mask) is not None and model.prefix_cache_hits == hits + 1
        model.prefix_cache = False
        assert fields_equal(changed, model.forward(px, ids3, mask)) == []
        model.prefix_cache = True
        # invalidation 3: new weights
        model._prefix_misses = 0
        model.forward(px, ids, mask)
        hits = model.prefix_cache_hits
        sd2 = synth.synth_state_dict(cfg, seed=1, lm_head=False)
        sd2["model.language_model.output.weight"] = sd["model.language_model.output.weight"]
        model.load_state_dict(sd2, strict=True)
        new_w = model.forward(px, ids, mask)
        assert model.prefix_cache_hits == hits and fields_equal(new_w, ref) != []
        hit = model.forward(px, ids, mask)
        assert model.prefix_cache_hits == hits + 1 and fields_equal(hit, new_w) == []
        # thrash guard: prompts that share no constant prefix (another one every forward) stop paying the prefix pass after three
        # misses in a row - those forwards run uncached - until a prefix repeats
        model._prefix, model._prefix_misses, model._prefix_last_miss = None, 0, None
        hits = model.prefix_cache_hits
        built = []
        for step in range(6):
            ids_v = ids.clone()
            ids_v[:, 5] += 1 + step
            before = model._prefix
            out_v = model.forward(px, ids_v, mask)
            built.append(model._prefix is not before)
            if step == 5:
                model.prefix_cache = False
                assert fields_equal(out_v, model.forward(px, ids_v, mask)) == []
                model.prefix_cache = True
        assert built == [True, True, True, False, False, False] and model.prefix_cache_hits == hits
        ids_v = ids.clone()
        ids_v[:, 5] += 6
        before = model._prefix
        model.forward(px, ids_v, mask)                      # the candidate of step 5 again: rebuilt ...
        assert model._prefix is not before and model.prefix_cache_hits == hits
        model.forward(px, ids_v, mask)                      # ... and served
        assert model.prefix_cache_hits == hits + 1 and model._prefix_misses == 0
    finally:
        model.use_gemm_workspace = True
    # with K-slicing on: re-association noise only (bound as in test_last_layer_trimming_is_invisible)
    model.prefix_cache = False
    plain = model.forward(px, ids, mask)
    model.prefix_cache = True
    model._prefix = None
    cold = model.forward(px, ids, mask)
    sliced = model.forward(px, ids, mask)
    assert fields_equal(cold, sliced) == []          # cold and warm cache: the same computation
    assert model._prefix["settings"][6] is True
    # (bounds derived, not fitted.  bf16: a cached and an uncached evaluation are two samples of the rounding noise the
    # reference's bf16 run carries against its fp32 run at these dims - 2B dims @224^2: the full_c1 fixture - and two samples differ
    # by sqrt(2) x one: bound 1.5 x the fixture's largest bf16-vs-fp32 distance of the row, the bound _full_case holds the HIP path
    # itself to.  mxfp8: a re-associated sum moves some FFN inputs across an e4m3 rounding boundary (2^-3 of the element, not 2^-8), so
    # two fp8 evaluations that slice K differently differ like two fp8 IMPLEMENTATIONS do: test_fp8_gpu.py holds those to 0.75 x the
    # oracles' fp8-vs-bf16 gap at these very dims (18.5 % / 19.7 % of the rows' norm: 0.139) - the same bound here.  Measured in round
    # 5: 2.9 % / 8.6 %; the values of every run are printed.)
    gnpz, gmeta = load_golden("full_c1")
    for f in ("hidden_state", "prompt_embedding"):
        e = rel_l2(getattr(sliced, f).float().cpu().numpy(), getattr(plain, f).float().cpu().numpy())
        own = max(rel_l2(gnpz[f"v{v['video_idx']}/{f}"], gnpz[f"v{v['video_idx']}/fp32/{f}"]) for v in gmeta["videos"]
                  if f"v{v['video_idx']}/fp32/{f}" in gnpz.files)
        bound = 1.5 * own if fmt == "bf16" else 0.75 * 0.185
        print(f"prefix cache on vs off, K-slicing on, {fmt}, {f}: relative L2 {e:.4f} (bound {bound:.4f}; the reference's own bf16-vs-fp32 {own:.4f})")
        assert e < bound, (f, e, bound)


def test_stressed_statistics_tiny_against_oracle(cuda):
    """trained-like statistics (synth.stress_tensors: 1 % of the hidden channels x 20 in the residual producers, attention logits
    of sigma 10, fc1 pre-activations of sigma 2 - the inputs on which the GELU fast path's vote fails and the optimistic softmax
    raises its offset in mid-sequence) at tiny dims against the ORACLE run on the same stressed weights: the kernels' rare paths
    are exercised end to end, not only by the kernel tests.  Both attention numerics; bounds relative to each field's magnitude,
    2 x what the benign tiny cases need (the stressed network amplifies rounding noise more: peaked softmaxes, |x| up to 100s)."""
    from mj_video_amd import synth
    from oracle import ref_cpu
    cfg = make_cfg("tiny", 56)
    sd = synth.synth_state_dict(cfg, seed=31)
    info = synth.stress_tensors(sd, cfg)
    vids = [dict(video_idx=0, n_tiles=4, caption_seed=1), dict(video_idx=1, n_tiles=3, caption_seed=2), dict(video_idx=2, n_tiles=4, caption_seed=3)]
    px, ids, mask, _ = case_inputs(cfg, vids, 9, 56)
    ref = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    worst = {}
    for scores in ("flash", "eager"):
        model = build_hip_model(cfg, sd, cuda)
        model.attention_scores = scores
        model.debug_probes = {}
        out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
        probes, model.debug_probes = model.debug_probes, None
        # the statistics really are the stressed ones: logits far beyond the benign +-2, pre-activations beyond the benign 0.64
        a0 = probes["llm_attn0"]
        qh = a0["q"].float()[:, :128]
        kh = a0["k"].float()[:, :128]
        assert (qh @ kh.t() / 128 ** 0.5).abs().max().item() > 15.0
        for f in ("score", "aspect_scores", "rewards", "aspect_gating_output", "aspect_weights", "criteria_gating_output", "hidden_state", "prompt_embedding"):
            r = ref[f].float()
            d = (getattr(out, f).float().cpu() - r).abs().max().item() / r.abs().max().item()
            worst[(scores, f)] = d
            assert torch.isfinite(getattr(out, f).float()).all(), f
    print("stressed tiny model vs oracle, max |d| / field magnitude:", {f"{k[0]}:{k[1]}": round(v, 4) for k, v in worst.items()}, info)
    for k, v in worst.items():
        assert v < 0.25, (k, v)


def test_k_sliced_path_is_no_further_from_fp32(cuda):
    """the pairwise bound above (K-sliced vs unsliced, 4.5 %) cannot tell a small systematic error of the K-sliced / skinny
    paths from re-association noise; the exact-integer kernel tests are the correctness gate, and this is the end-to-end
    companion: on the full_c1 fixture (which holds the reference's fp32 run) the forward WITH the split-K scratch must be no
    further from the fp32 hidden rows than the forward without it (up to the spread two samples of one noise show), and the
    measured distances are printed next to the reference's own bf16-vs-fp32 distance."""
    from mj_video_amd import synth
    npz, meta = load_golden("full_c1")
    cfg = make_cfg("2b", 224)
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    vids = [v for v in meta["videos"] if f"v{v['video_idx']}/fp32/hidden_state" in npz.files]
    assert vids
    px, ids, mask, _ = case_inputs(cfg, vids, meta["pixel_seed"], 224)
    outs = {}
    try:
        for use_ws in (True, False):
            model.use_gemm_workspace = use_ws
            outs[use_ws] = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    finally:
        model.use_gemm_workspace = True
    for f in ("hidden_state", "prompt_embedding"):
        f32 = np.stack([npz[f"v{v['video_idx']}/fp32/{f}"] for v in vids]).astype(np.float32).reshape(len(vids), -1)
        b16 = np.stack([npz[f"v{v['video_idx']}/{f}"] for v in vids]).astype(np.float32).reshape(len(vids), -1)
        d_ref = rel_l2(b16, f32)
        d_on = rel_l2(getattr(outs[True], f).float().cpu().numpy().reshape(len(vids), -1), f32)
        d_off = rel_l2(getattr(outs[False], f).float().cpu().numpy().reshape(len(vids), -1), f32)
        print(f"{f}: distance to the fp32 run - reference bf16 {d_ref:.4f}, HIP with K slicing {d_on:.4f}, without {d_off:.4f}")
        assert d_on <= 1.3 * d_off + 3e-3, (f, d_on, d_off)
        assert d_on <= 1.5 * d_ref and d_off <= 1.5 * d_ref, (f, d_on, d_off, d_ref)


@pytest.mark.parametrize("scores,fused", [("flash", True), ("eager", False), ("flash", False)])
@pytest.mark.parametrize("name", ["vit_layer0", "vit_layer23", "llm_layer0", "llm_layer23"])
def test_single_layer_at_production_shape(cuda, name, scores, fused):
    """tests/golden/layers.npz: ONE layer at MJ-VIDEO-2B dimensions and the headline sequence lengths, executed by the
    reference's own InternVisionEncoderLayer.forward (modeling_intern_vit.py:283-295, [2, 1025, 1024]) /
    InternLM2DecoderLayer.forward (modeling_internlm2.py:621-681, [1, 2186, 2048]) on seed-defined bf16 rows with the
    synthetic weights of that layer - no compounding over 24 layers, so the bound can be tight: relative L2 of the
    sampled output rows against the reference's bf16 run <= 2 x the reference's own bf16-vs-fp32 distance for this
    layer (0.30 % vision, 0.44 % language), and the HIP rows no further from the fp32 run than 2 x that either."""
    from util import layer_input_rows, layer_tensors
    from mj_video_amd import synth
    from mj_video_amd.modeling import InternVLChatRewardModeling
    npz, meta = load_golden("layers")
    m = meta["layers"]
    case = next(c for c in m["cases"] if c["name"] == name)
    cfg = make_cfg("2b", m["image_size"])
    full_prefix = (f"model.vision_model.encoder.layers.{case['layer']}." if case["tower"] == "vit"
                   else f"model.language_model.model.layers.{case['layer']}.")
    w = layer_tensors(cfg, full_prefix, m["weight_seed"])
    cfg.vision_config.num_hidden_layers = 1      # a one-layer skeleton per tower: the layer under test sits at index 0
    cfg.llm_config.num_hidden_layers = 1
    cfg.llm_config.vocab_size = 128
    model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16)
    for prm in model.parameters():
        prm.data.zero_()
    mod = model.model.vision_model.encoder.layers[0] if case["tower"] == "vit" else model.model.language_model.model.layers[0]
    mod.load_state_dict(w, strict=True)
    from util import apply_test_overrides
    model = apply_test_overrides(model.to(torch.bfloat16).to(cuda).eval())
    model.attention_scores = scores     # both of the reference's attention numerics are held to the same bound
    model.norm_fusion = fused           # ... and the norms folded into their GEMMs (opt-in) as well as the reference's rounding points (default)
    x = layer_input_rows(m["input_seed"], case["input_tag"], tuple(case["shape"]))
    y = (model.run_vit_layer(0, x) if case["tower"] == "vit" else model.run_llm_layer(0, x)).float().cpu()
    rows = npz[f"{name}/rows"]
    got = y[:, rows].numpy()
    ref = bits_to_f32(npz[f"{name}/out"])
    f32 = npz[f"{name}/fp32"]
    noise = case["ref_bf16_vs_fp32"]
    d_ref, d_f32 = rel_l2(got, ref), rel_l2(got, f32)
    print(f"{name} [{scores} scores, norms {'folded' if fused else 'separate'}]: HIP vs reference bf16 {d_ref:.5f}, HIP vs reference fp32 {d_f32:.5f}; reference bf16 vs fp32 {noise:.5f}")
    assert np.isfinite(got).all()
    assert d_ref <= 2.0 * noise, (name, d_ref, noise)
    assert d_f32 <= 2.0 * noise, (name, d_f32, noise)


def test_sticky_dynamic_ntk_sequence(cuda):
    """the reference's rotary cache is stateful (modeling_internlm2.py:169-176,204-229): ONE model scored short -> long ->
    short -> a padded batch with max_position_embeddings (48) below every sequence length and dynamic-NTK scaling; the
    third call must reproduce the reference's THIRD call (rescaled base still in force), not its first, and the padded
    batch regrows the cache by its PADDED width.  Bounds: the tiny-dims noise floor of these very calls."""
    from mj_video_amd import configuration as C, synth
    npz, meta = load_golden("layers")
    m = meta["ntk"]
    cd = C.tiny_config_dict(m["image_size"])
    cd["llm_config"]["max_position_embeddings"] = m["max_position_embeddings"]
    cd["llm_config"]["rope_scaling"] = dict(m["rope_scaling"])
    cfg = C.InternVLChatRewardModelingConfig(**cd, **C.mjvideo_head_kwargs())
    model = build_hip_model(cfg, synth.synth_state_dict(cfg, seed=m["weight_seed"]), cuda)
    names = [f"ntk/{c['name']}" for c in m["calls"]]
    outs = {}
    for call in m["calls"]:
        px, ids, mask, _ = case_inputs(cfg, call["videos"], m["pixel_seed"], m["image_size"])
        out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
        outs[call["name"]] = out
        for f in FIELDS:
            got = getattr(out, f).float().cpu().numpy()
            ref = npz[f"ntk/{call['name']}/{f}"]
            tol = TOL_FACTOR * noise_floor(npz, names, f) + ATOL_FLOOR
            assert np.abs(got - ref).max() <= tol, (call["name"], f, float(np.abs(got - ref).max()), tol)
    # the state is real: the third call is closer to the reference's third call than to its first
    h3 = outs["short_again"].hidden_state.float().cpu().numpy()
    d_third = np.abs(h3 - npz["ntk/short_again/hidden_state"]).max()
    d_first = np.abs(h3 - npz["ntk/short_first/hidden_state"]).max()
    print(f"third call: max |d| to the reference's third call {d_third:.4f}, to its first call {d_first:.4f}")
    # (reported only: at tiny dims the reference's own first and third calls differ by 0.035 at most, 0.009 rms - about the
    # bf16 noise of these rows - so which of the two a noisy output lands nearer to is a coin toss.  The state itself decides:)
    assert not torch.equal(outs["short_first"].hidden_state, outs["short_again"].hidden_state)
    # the tables the model would use for a short call NOW are the oracle's tables after the same call sequence, bit for bit,
    # and not the tables of a freshly constructed model
    from oracle import ref_cpu
    state = {}
    widths = []
    for call in m["calls"]:
        _, ids, _, _ = case_inputs(cfg, call["videos"], m["pixel_seed"], m["image_size"])
        widths.append(ids.shape[1])
        ref_cpu.rope_tables(cfg, ids.shape[1], torch.bfloat16, state)
    assert model._rope_state["cached"] == state["cached"] == max(widths) and model._rope_state["base"] == state["base"]
    assert state["base"] > float(cfg.llm_config.rope_theta)
    n = widths[0]
    cos_m, sin_m = model._rope_tables(n, cuda)
    cos_o, sin_o = ref_cpu.rope_tables(cfg, n, torch.bfloat16, state)
    cos_f, _ = ref_cpu.rope_tables(cfg, n, torch.bfloat16)
    assert torch.equal(cos_m[:n].cpu().view(cos_o.shape), cos_o) and torch.equal(sin_m[:n].cpu().view(sin_o.shape), sin_o)
    assert not torch.equal(cos_o, cos_f)


def test_two_threads_two_streams_score_bitwise(cuda):
    """ABI 4 keeps no setting between calls (tile / attention kernel / split-K scratch travel in the descriptors; the Python
    wrappers keep their defaults per thread): two threads, each with its own model instance (same weights) and its own HIP
    stream, score different batches at the same time - at MJ-VIDEO-2B dims @224^2, so the K-sliced GEMM tails (workspace)
    and every tile kernel are on the path - and every field equals the single-threaded result bit for bit."""
    import threading
    from mj_video_amd import synth
    cfg = make_cfg("2b", 224)
    sd = synth.synth_state_dict(cfg, seed=0, lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    models = [build_hip_model(cfg, sd, cuda) for _ in range(2)]
    batches = []
    for b, tiles in enumerate(([8, 6, 8], [5, 8, 3, 8])):
        vids = [dict(video_idx=10 * b + i, n_tiles=t, caption_seed=10 * b + i) for i, t in enumerate(tiles)]
        px, ids, mask, _ = case_inputs(cfg, vids, 90 + b, 224)
        batches.append((px.to(cuda), ids.to(cuda), mask.to(cuda)))
    expect = [models[i].forward(*batches[i]) for i in range(2)]
    torch.cuda.synchronize()
    for rnd in range(3):
        got, errs = [None, None], []

        def work(i):
            try:
                st = torch.cuda.Stream(device=cuda)
                with torch.cuda.stream(st):
                    got[i] = models[i].forward(*batches[i])
                st.synchronize()
            except Exception as e:   # surfaced below
                errs.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        for i in range(2):
            for f in FIELDS:
                assert torch.equal(getattr(got[i], f), getattr(expect[i], f)), (rnd, i, f)


def test_forward_is_deterministic_at_headline_shape(cuda):
    """headline batch (8 videos x 8 tiles @448: M = 65 600 / 17 488 - every large GEMM peels tail rows, the long-K tails split
    K, the ViT attention runs its one-wave ragged block): repeated forwards must agree bit for bit (split-K sums its slices
    in a fixed order, no atomics anywhere), which also screens the kernels for races under a full-size load"""
    from mj_video_amd import synth
    cfg = make_cfg("2b", 448)
    sd = synth.synth_state_dict(cfg, seed=0, lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    vids = [dict(video_idx=i, n_tiles=8, caption_seed=i) for i in range(8)]
    px, ids, mask, _ = case_inputs(cfg, vids, 78, 448)
    px, ids, mask = px.to(cuda), ids.to(cuda), mask.to(cuda)
    ref = model.forward(px, ids, mask)
    torch.cuda.synchronize()
    assert torch.isfinite(ref.score).all()
    for it in range(4):
        got = model.forward(px, ids, mask)
        torch.cuda.synchronize()
        for f in FIELDS:
            assert torch.equal(getattr(got, f), getattr(ref, f)), (it, f)


def test_error_behaviour(cuda):
    """the reference's ValueErrors (moe_reward.py:57,218-219) and the build's loud failures"""
    from mj_video_amd import synth
    cfg = make_cfg("tiny", 56)
    sd = synth.synth_state_dict(cfg, seed=3, dtype=torch.float32)
    model = build_hip_model(cfg, sd, cuda)
    px = synth.synth_pixel_values(1, 0, 2, 56).to(cuda)
    ids = synth.synth_input_ids(8, 1)
    bad = ids.clone()
    bad[0, -1] = 5  # break the gating pattern
    with pytest.raises(ValueError, match="Token pattern not found"):
        model.forward(px, bad.to(cuda), torch.ones_like(bad).to(cuda))
    model.config.pad_token_id = None
    with pytest.raises(ValueError, match="Cannot handle batch sizes > 1"):
        model.forward(torch.cat([px, px]), torch.cat([ids, ids]).to(cuda), None)
    model.config.pad_token_id = synth.PAD_ID
    with pytest.raises(ValueError, match="IMG_CONTEXT"):
        model.forward(px[:1], ids.to(cuda), None)
    with pytest.raises(TypeError):
        model.forward(px.float(), ids.to(cuda), None)
    # attention_mask=None and pad_token_id=None with batch 1 both work and agree
    a = model.forward(px, ids.to(cuda), None).score.item()
    model.config.pad_token_id = None
    b = model.forward(px, ids.to(cuda), torch.ones_like(ids).to(cuda)).score.item()
    assert a == b


def test_collated_batch_layout(cuda):
    """the training collator's batch layout ([B,F,3,H,W] pixels, right-padded ids, two videos per pair) scored in one
    packed forward == the two videos scored separately"""
    from mj_video_amd import harness, synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    cfg = make_cfg("tiny", 56)
    model = build_hip_model(cfg, synth.synth_state_dict(cfg, seed=9, dtype=torch.float32), cuda)
    B, Fr = 3, 4
    per = num_image_tokens_per_tile(cfg) * Fr
    batch = {}
    for i in (0, 1):
        batch[f"video_{i}_pixel_values"] = torch.stack([synth.synth_pixel_values(5, 10 * i + b, Fr, 56) for b in range(B)])
        ids, mask = synth.pad_batch([synth.synth_input_ids(per, 20 + b, n_caption=32 - 3 * b * i) for b in range(B)], length=per + 150)
        batch[f"video_{i}_input_ids"], batch[f"video_{i}_attention_mask"] = ids, mask
    o0, o1 = harness.score_collated_batch(model, batch)
    from oracle import ref_cpu
    tiny_npz, tiny_meta = load_golden("tiny")
    names = [c["name"] for c in tiny_meta["cases"]]
    sd = synth.synth_state_dict(cfg, seed=9, dtype=torch.float32)
    sd_bf = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    for i, o in ((0, o0), (1, o1)):
        ref = model.forward(batch[f"video_{i}_pixel_values"].reshape(-1, 3, 56, 56).to(cuda), batch[f"video_{i}_input_ids"].to(cuda),
                            batch[f"video_{i}_attention_mask"].to(cuda))
        for f in FIELDS:
            assert torch.equal(getattr(o, f), getattr(ref, f)), (i, f)
        # ... and against the ORACLE on the collated (right-padded, batch-3) layout itself
        orc = ref_cpu.reward_forward(sd_bf, cfg, batch[f"video_{i}_pixel_values"].reshape(-1, 3, 56, 56),
                                     batch[f"video_{i}_input_ids"], batch[f"video_{i}_attention_mask"],
                                     synth.IMG_CONTEXT_ID, synth.PAD_ID)
        for f in ("score", "aspect_scores", "rewards", "aspect_gating_output", "aspect_weights", "criteria_gating_output"):
            tol = TOL_FACTOR * noise_floor(tiny_npz, names, f) + ATOL_FLOOR
            dlt = float((getattr(o, f).float().cpu() - orc[f].float()).abs().max())
            assert dlt <= tol, (i, f, dlt, tol)


def test_random_batch_compositions_against_the_oracle(cuda):
    """tiny dims, 12 random batches - 1 to 4 videos, 1 to 5 tiles each, caption lengths from 1 to 60 tokens (so right-padded
    rows of very different lengths, prompts whose gating token sits at different offsets), interleaved or blocked frame
    prefixes - scored by the HIP path and by the oracle (bf16, CPU) on the very same tensors.  Every head output within the
    tiny-dims noise bound of the reference-executed fixtures; fresh weights per batch."""
    import random
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    from oracle import ref_cpu
    tiny_npz, tiny_meta = load_golden("tiny")
    names = [c["name"] for c in tiny_meta["cases"]]
    rng = random.Random(31)
    cfg = make_cfg("tiny", 56)
    per = num_image_tokens_per_tile(cfg)
    worst = {}
    for it in range(12):
        sd = synth.synth_state_dict(cfg, seed=100 + it, dtype=torch.float32)
        model = build_hip_model(cfg, sd, cuda)
        sd_bf = {k: v.to(torch.bfloat16) for k, v in sd.items()}
        B = rng.randint(1, 4)
        tiles = [rng.randint(1, 5) for _ in range(B)]
        px = torch.cat([synth.synth_pixel_values(200 + it, b, tiles[b], 56) for b in range(B)])
        rows = [synth.synth_input_ids(per * tiles[b], 300 + 10 * it + b, n_caption=rng.randint(1, 60),
                                      interleave_frames=(tiles[b] if rng.random() < 0.5 else None)) for b in range(B)]
        ids, mask = synth.pad_batch(rows, length=max(int(r.shape[-1]) for r in rows) + rng.choice([0, 0, 7, 40]))
        out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
        orc = ref_cpu.reward_forward(sd_bf, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
        for f in ("score", "aspect_scores", "rewards", "aspect_gating_output", "aspect_weights", "criteria_gating_output",
                  "hidden_state", "prompt_embedding"):
            got = getattr(out, f).float().cpu()
            assert torch.isfinite(got).all(), (it, f)
            tol = TOL_FACTOR * noise_floor(tiny_npz, names, f) + ATOL_FLOOR
            dlt = float((got - orc[f].float()).abs().max())
            worst[f] = max(worst.get(f, 0.0), dlt / tol)
            assert dlt <= tol, (it, B, tiles, f, dlt, tol)
    print("random batches: worst deviation / tolerance per field", {k: round(v, 3) for k, v in worst.items()})


def test_eval_driver_protocol(cuda):
    """scripts/eval/eval_genai_mjvideo.py: batched scoring of (caption, left, right, vote) examples through
    prepare_chat_input + device preprocessing gives the same scores as the reference's protocol (one forward per video,
    batch 1) and the same prefer_Acc / Acc bookkeeping"""
    import importlib.util
    import numpy as np
    from util import ROOT
    from mj_video_amd import harness, synth, video
    from mj_video_amd.chat_input import prepare_chat_input, video_prefix
    from test_host_logic import StubTokenizer
    spec = importlib.util.spec_from_file_location("eval_driver", f"{ROOT}/scripts/eval/eval_genai_mjvideo.py")
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    cfg = make_cfg("tiny", 56)
    model = build_hip_model(cfg, synth.synth_state_dict(cfg, seed=31, dtype=torch.float32), cuda)
    tok = StubTokenizer()
    rng = np.random.default_rng(5)
    store = {f"v{i}": rng.integers(0, 256, size=(4, 90, 120, 3), dtype=np.uint8) for i in range(10)}
    votes = ["leftvote", "rightvote", "tievote", "bothbad_vote", "rightvote"]
    examples = [dict(prompt=f"a synthetic caption number {i}", left_video=f"v{2 * i}", right_video=f"v{2 * i + 1}",
                     vote_type=votes[i]) for i in range(5)]

    def loader(name):
        return video.load_frames_device(torch.from_numpy(store[name]).to(cuda), input_size=56, max_num=1)[0]

    counts, scores = drv.evaluate_examples(model, cfg, tok, examples, loader, pairs_per_batch=2)
    ref = harness.PreferenceCounts()
    for i, ex in enumerate(examples):
        s = []
        for side in ("left_video", "right_video"):
            pv = loader(ex[side])
            ids, mask = prepare_chat_input(cfg, tok, pv, video_prefix(pv.shape[0]) + ex["prompt"], {}, device=cuda)
            s.append(model.forward(pv, ids, mask).score[0].item())
        assert s[0] == scores[i, 0, 0].item() and s[1] == scores[i, 1, 0].item()
        ref.update(ex["vote_type"], s[0], s[1])
    assert (counts.prefer_truth, counts.prefer_total, counts.truth, counts.total) == \
        (ref.prefer_truth, ref.prefer_total, ref.truth, ref.total)
    assert counts.total == 5 and counts.prefer_total == 3
    # the prefetching form (decoded frames in host memory, uploads of batch i + 1 on a copy stream while batch i is scored):
    # bit-identical scores, same counts
    counts2, scores2 = drv.evaluate_examples(model, cfg, tok, examples, None, pairs_per_batch=2,
                                             host_loader=lambda name: torch.from_numpy(store[name]),
                                             to_device_pixels=lambda t: video.load_frames_device(t, input_size=56, max_num=1)[0])
    assert torch.equal(scores2, scores)
    assert (counts2.prefer_truth, counts2.truth, counts2.total) == (counts.prefer_truth, counts.truth, counts.total)


def test_prefetch_to_device_order_values_and_nesting(cuda):
    """harness.prefetch_to_device: every item arrives, in order, bit for bit, whatever the nesting (tensor / tuple / dict),
    usable on the current stream straight away, with one or two uploads in flight"""
    from mj_video_amd import harness
    g = torch.Generator().manual_seed(1)
    items = [dict(a=torch.randn(3, 1000, generator=g), b=(torch.randint(0, 255, (2, 777), dtype=torch.uint8, generator=g), i))
             for i in range(7)]
    for depth in (1, 2):
        got = []
        for it in harness.prefetch_to_device(iter(items), cuda, depth=depth):
            assert it["a"].is_cuda and it["b"][0].is_cuda
            got.append((it["a"] * 2.0, it["b"][0].clone(), it["b"][1]))      # consume on the current stream immediately
        torch.cuda.synchronize()
        assert len(got) == 7
        for i, (a2, b, tag) in enumerate(got):
            assert tag == i and torch.equal(a2.cpu(), items[i]["a"] * 2.0) and torch.equal(b.cpu(), items[i]["b"][0])


def test_mjbench_video_driver(cuda):
    """scripts/eval/eval_mjbench_video.py: pairs in the datas/test.json schema scored in packed batches through
    prepare_chat_input + device preprocessing give the same [pairs, 2, 34] block as one forward per video (the reference's
    loop, overall_train.py:407-421), and the metrics are harness.evaluate_mjbench of that block (the bookkeeping itself is
    held to the reference's own methods on the CPU: tests/test_host_fixtures.py)"""
    import importlib.util
    import sys as _sys
    import numpy as np
    from util import GOLDEN, ROOT
    from mj_video_amd import harness, synth, video
    from mj_video_amd.chat_input import prepare_chat_input, video_prefix
    from test_host_logic import StubTokenizer
    _sys.path.insert(0, GOLDEN)
    import make_mjbench_fixture as gen
    spec = importlib.util.spec_from_file_location("mjbench_driver", f"{ROOT}/scripts/eval/eval_mjbench_video.py")
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    cfg = make_cfg("tiny", 56)
    model = build_hip_model(cfg, synth.synth_state_dict(cfg, seed=31, dtype=torch.float32), cuda)
    assert model.num_aspects == 5 and model.num_objectives == 28
    tok = StubTokenizer()
    rng = np.random.default_rng(7)
    items = gen.synth_items(9, 5)
    store = {}
    for it in items:
        for v in (0, 1):
            store[os.path.join("/videos", it[f"video_{v}_path"])] = rng.integers(0, 256, size=(2, 90, 120, 3), dtype=np.uint8)

    def loader(path):
        return video.load_frames_device(torch.from_numpy(store[path]).to(cuda), input_size=56, max_num=1)[0]

    metrics, scores = drv.evaluate_items(model, cfg, tok, items, loader, "/videos", pairs_per_batch=2)
    for i, it in enumerate(items):
        for v in (0, 1):
            pv = loader(os.path.join("/videos", it[f"video_{v}_path"]))
            ids, mask = prepare_chat_input(cfg, tok, pv, video_prefix(pv.shape[0]) + it["caption"], {}, device=cuda)
            out = model.forward(pv, ids, mask)
            one = torch.cat([out.score.float().view(1), out.aspect_scores.float().view(-1), out.rewards.float().view(-1)])
            assert torch.equal(one, scores[i, v]), (i, v)
    again = harness.evaluate_mjbench(items, scores.cpu().numpy())
    assert {k: metrics[k] for k in ("overall_accuracy", "overall_correct", "overall_count")} == \
        {k: again[k] for k in ("overall_accuracy", "overall_correct", "overall_count")}
    assert metrics["aspect"]["tp"] == again["aspect"]["tp"] and metrics["criteria"]["fn"] == again["criteria"]["fn"]
    assert metrics["overall_count"] == sum(it["overall_preference"] in ("Video 1 better", "Video 2 better") for it in items)


def _small_backbone_2b_heads(cuda, image_size=224):
    """MJ-VIDEO-2B head dimensions on a one-layer backbone (the backbone is never run by heads_forward)"""
    from mj_video_amd import synth
    from mj_video_amd.modeling import InternVLChatRewardModeling
    cfg = make_cfg("2b", image_size)
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 1
    cfg.llm_config.vocab_size = 128
    model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16)
    for prm in model.parameters():
        prm.data.zero_()
    return cfg, model


def test_reward_heads_kernel_against_oracle(cuda):
    """reward_heads kernel + the gating GEMMs in isolation (heads.hip; moe_reward.py:239-297): given the SAME two hidden
    rows per sample, every head output against the oracle's head math (proven bit-identical to the reference's own head
    code by make_golden.gen_rankhid / gen_rankset_eng).  bf16 fields: <= 2 ulps (fp32 accumulation order of the matvecs
    and of the three 1024-wide gating layers differs; each layer boundary is one bf16 rounding) or 1e-3 absolute where a
    value sits next to zero; the fp32 score / aspect_scores: 1 % of the magnitude + 2e-3."""
    from mj_video_amd import synth
    from oracle import ref_cpu
    from util import bf16_ulps
    cfg, model = _small_backbone_2b_heads(cuda)
    head_sd = synth.synth_head_state_dict(cfg, seed=5)
    assert not model.load_state_dict(head_sd, strict=False).unexpected_keys
    model = model.to(cuda).eval()
    g = torch.Generator().manual_seed(11)
    B, H = 7, cfg.llm_config.hidden_size
    h_r = (torch.randn(B, H, generator=g) * 1.1).to(torch.bfloat16)
    h_g = (torch.randn(B, H, generator=g) * 0.9).to(torch.bfloat16)
    out = model.heads_forward(h_r.to(cuda), h_g.to(cuda))
    torch.cuda.synchronize()
    ref = ref_cpu.reward_heads(head_sd, cfg, h_r, h_g)
    worst = {}
    for f in FIELDS:
        got, exp = getattr(out, f).float().cpu(), ref[f].float()
        assert got.shape == exp.shape, (f, got.shape, exp.shape)
        if getattr(out, f).dtype == torch.bfloat16:
            ul = bf16_ulps(got, exp)
            ok = (ul <= 2) | ((got - exp).abs() <= 1e-3)
            worst[f] = int(ul.max())
            assert ok.all(), (f, int(ul.max()), float((got - exp).abs().max()))
        else:
            err = (got - exp).abs()
            worst[f] = float(err.max())
            assert (err <= 0.01 * exp.abs() + 2e-3).all(), (f, float(err.max()))
    assert torch.equal(out.hidden_state.cpu(), h_r) and torch.equal(out.prompt_embedding.cpu(), h_g)
    print("heads vs oracle, worst (ulps for bf16 fields, abs for fp32):", worst)


def _c2_pairs(cuda, n_pairs):
    """model + inputs of the first n_pairs pairs of the rank set at the headline shape (8 frames @448, N = 2186)"""
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    npz, meta = load_golden("rankset_c2")
    cfg = make_cfg("2b", meta["image_size"])
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    nt = meta["n_tiles"]
    pairs = []
    for p in range(n_pairs):
        row = synth.synth_input_ids(num_image_tokens_per_tile(cfg) * nt, caption_seed=meta["caption_seed_base"] + p)
        pairs.append((row, [synth.synth_pixel_values(meta["pixel_seed"], 2 * p + j, nt, meta["image_size"]) for j in range(2)]))
    return model, cfg, npz, meta, pairs


def _score_pairs(model, cuda, pairs):
    from mj_video_amd import synth
    px, ids = [], []
    for row, vids in pairs:
        px += vids
        ids += [row, row]
    ids_b, mask = synth.pad_batch(ids)
    model.forward(torch.cat(px).to(cuda), ids_b.to(cuda), mask.to(cuda))
    return model.last_packed34.float().reshape(-1, 2, 34)


def test_c3_shard_of_16_videos_and_rccl_allgather(cuda):
    """BASELINE.json configs[2]: 64 pairs over 8 GPUs = 8 pairs = 16 videos per GPU.  (a) one rank's shard in ONE
    packed forward (M = 131 200 ViT rows, 34 976 LLM rows) against the reference's scores of those 16 videos, within the
    bounds the 4-pair batches of the rank-set test meet; (b) the N > 1 code path of bench.py / parallel.score_pairs_dp on
    the RCCL backend with world_size 1: the all_gather_into_tensor of the [pairs, 2, 34] block returns the shard unchanged."""
    import os
    import torch.distributed as dist
    from mj_video_amd import parallel
    model, cfg, npz, meta, pairs = _c2_pairs(cuda, 8)
    got = _score_pairs(model, cuda, pairs).cpu().numpy()
    ref = npz["ref_bf16"][:8]
    assert np.isfinite(got).all()
    d = np.abs(got[..., 0] - ref[..., 0])
    noise_rms = pooled_noise_rms("score")
    print(f"16-video shard: |hip - ref| score rms {np.sqrt((d ** 2).mean()):.4f} max {d.max():.4f}; reference noise rms {noise_rms:.4f}")
    # rms of only 16 deviations: the bound of the rank-set test (2 x the reference's own noise rms, asserted there on 512
    # scores) widened by the 99.9 % quantile of a 16-sample rms, sqrt(chi2_16(0.999) / 16) = 1.57 - round 3 measured the
    # same 16 videos at rms 0.100 (round-2 attention kernel) and 0.144 (round-3 kernel) while the 512-score rms moved
    # 0.102 -> 0.112: a 16-sample rms scatters by +-18 % (1 sigma) under any re-association of the fp32 sums
    from scipy.stats import chi2
    assert np.sqrt((d ** 2).mean()) <= 2.0 * noise_rms * np.sqrt(chi2.ppf(0.999, d.size) / d.size) + ATOL_FLOOR
    assert d.max() <= TOL_FACTOR * pooled_noise_floor("score") + ATOL_FLOOR
    # the same videos in two 8-video forwards: equal up to fp32 summation order (tile / split-K choice follows M)
    two = torch.cat([_score_pairs(model, cuda, pairs[:4]), _score_pairs(model, cuda, pairs[4:])]).cpu().numpy()
    assert np.abs(two[..., 0] - got[..., 0]).max() <= TOL_FACTOR * pooled_noise_floor("score") + ATOL_FLOOR
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        calls = []

        def score_fn(local):
            calls.append(len(local))
            return torch.cat([_score_pairs(model, cuda, local[i:i + 4]) for i in range(0, len(local), 4)])

        full = parallel.score_pairs_dp(score_fn, pairs, device=cuda)
        assert calls == [8] and full.shape == (8, 2, 34) and full.is_cuda
        assert torch.equal(full.cpu(), torch.from_numpy(two))
    finally:
        if created:
            dist.destroy_process_group()


def test_c4lite_against_golden(cuda):
    """BASELINE.json configs[3] (16 frames x dynamic tiles, long-context image tokens) at the largest size the eager
    reference could run: 48 tiles, N = 12 426 tokens in ONE sequence (tests/golden/c4lite.npz, reference bf16 only).
    Hidden rows within LAYER_TOL of the reference; head outputs within the pooled noise bounds of the 2 186-token
    fixtures (no fp32 run exists at this size)."""
    from mj_video_amd import synth
    try:
        npz, meta = load_golden("c4lite")
    except FileNotFoundError:
        pytest.skip("c4lite fixture not generated")
    cfg = make_cfg("2b", meta["image_size"])
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    vids = [dict(video_idx=0, n_tiles=meta["n_tiles"], caption_seed=meta["caption_seed"])]
    px, ids, mask, _ = case_inputs(cfg, vids, meta["pixel_seed"], meta["image_size"])
    assert ids.shape[1] == meta["seq_len"]
    out = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    for f in FIELDS:
        got, ref = getattr(out, f)[0].float().cpu().numpy(), npz[f"v0/{f}"][0]
        assert np.isfinite(got).all(), f
        if f in ("hidden_state", "prompt_embedding"):
            e = rel_l2(got, ref)
            print(f"c4lite {f}: relative L2 {e:.4f}")
            assert e < LAYER_TOL["llm"], (f, e)
            continue
        tol = TOL_FACTOR * pooled_noise_floor(f) + ATOL_FLOOR if f in _PACKED34 else \
            TOL_FACTOR * max(noise_floor(*_c2_noise(), f), 1e-3) + ATOL_FLOOR
        d = float(np.abs(got - ref).max())
        print(f"c4lite {f}: max|d|={d:.3e} tol={tol:.3e}")
        assert d <= tol, (f, d, tol)
    again = model.forward(px.to(cuda), ids.to(cuda), mask.to(cuda))
    for f in FIELDS:
        assert torch.equal(getattr(out, f), getattr(again, f)), f


def _c2_noise():
    npz, meta = load_golden("full_c2")
    return npz, [f"v{v['video_idx']}" for v in meta["videos"] if f"v{v['video_idx']}/fp32/score" in npz.files]


def test_c4_full_112_tiles_end_to_end(cuda):
    """BASELINE.json configs[3] at full size: 16 frames x 7 tiles = 112 tiles, N = 28 810 tokens in one sequence (the eager
    reference cannot run it: 26.6 GB of scores + a 53 GB fp32 copy per layer).  The whole forward is finite and bit-
    deterministic, and the first decoder layer's causal GQA attention - on the q / k / v the real forward produced - agrees
    with a chunked fp32 evaluation with the reference's score rounding (modeling_internlm2.py:383-411)."""
    from mj_video_amd import synth
    cfg = make_cfg("2b", 448)
    sd = synth.synth_state_dict(cfg, seed=0, lm_head=False)
    sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(
        cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
    model = build_hip_model(cfg, sd, cuda)
    vids = [dict(video_idx=0, n_tiles=112, caption_seed=5)]
    px, ids, mask, _ = case_inputs(cfg, vids, 601, 448)
    N = ids.shape[1]
    assert N == 112 * 256 + 138
    px, ids, mask = px.to(cuda), ids.to(cuda), mask.to(cuda)
    a = model.forward(px, ids, mask)
    b = model.forward(px, ids, mask)
    torch.cuda.synchronize()
    for f in FIELDS:
        assert torch.isfinite(getattr(a, f).float()).all(), f
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    model.debug_probes = {}
    c = model.forward(px, ids, mask)     # untrimmed last layer, every layer's state cloned
    torch.cuda.synchronize()
    pr = model.debug_probes["llm_attn0"]
    model.debug_probes = None
    assert abs(c.score.item() - a.score.item()) <= TOL_FACTOR * pooled_noise_floor("score") + ATOL_FLOOR
    q, k, v, o = pr["q"].float(), pr["k"].float(), pr["v"].float(), pr["out"].float()
    D, KV = 128, pr["kv_heads"]
    G = q.shape[1] // (KV * D)
    rows = torch.cat([torch.arange(0, 64), torch.arange(N // 2, N // 2 + 96), torch.arange(N - 96, N)]).to(cuda)
    scale = D ** -0.5
    for head in (0, G * KV - 1):
        kvh = head // G
        kh = k[:, kvh * D:(kvh + 1) * D]
        vh = v[:, kvh * (G + 2) * D:kvh * (G + 2) * D + D] if v.shape[1] != KV * D else v[:, kvh * D:(kvh + 1) * D]
        sc = q[rows, head * D:(head + 1) * D] @ kh.t()
        sc = (sc.to(torch.bfloat16).float() * scale).to(torch.bfloat16).float()
        sc = sc.masked_fill(torch.arange(N, device=cuda)[None, :] > rows[:, None], float("-inf"))
        ref = (torch.softmax(sc, -1).to(torch.bfloat16).float() @ vh).to(torch.bfloat16).float()
        got = o[rows, head * D:(head + 1) * D]
        rel = ((got - ref).norm() / ref.norm()).item()
        print(f"C4 layer-0 attention, head {head}: relative L2 vs chunked fp32 {rel:.2e}")
        assert rel < 6e-3 and (got - ref).abs().max().item() < 0.03


def test_bench_launcher_path_runs_on_one_gpu(cuda):
    """The N > 1 path of bench.py - child `python -m torch.distributed.run`, RCCL process group, score_pairs_dp's all-gather,
    ranks_seen - executed at world size 1 (MJV_BENCH_FORCE_LAUNCHER=1): the code the driver's 8-GPU run goes through must have
    run somewhere before it.  The bench is started as a CHILD process (never exec'd from this one)."""
    import subprocess
    import sys
    env = dict(os.environ, MJV_BENCH_FORCE_LAUNCHER="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MJV_LIBRARY", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "2",
                        "--no-cpu-baseline", "--no-latency"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["process_group"] == "nccl" and line["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and line["config"]["global_pairs_per_step"] == 2
    assert line["library"] == {"path": os.path.join("mj-video_amd", "libmjv_hip.so"), "bench_build": False}
