"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors the REFERENCE produced (tests/golden, make_golden.py).

The fixtures were generated in this image by running the reference's own forward; the oracle uses the same
torch ops in the same order, so on the same CPU class it must reproduce them BIT FOR BIT.  (On a host whose
oneDNN picks different bf16 GEMM kernels the last bit may differ; then half the reference's own bf16-vs-fp32
noise floor is the bound.)"""
import numpy as np
import pytest
import torch

from util import FIELDS, case_inputs, case_inputs_masked, load_golden, make_cfg
from mj_video_amd import synth
from oracle import ref_cpu


def _check(out, npz, prefix, names):
    exact = True
    for f in FIELDS:
        got = out[f].float().numpy()
        ref = npz[f"{prefix}/{f}"]
        assert got.shape == ref.shape, (prefix, f)
        if np.array_equal(got, ref):
            continue
        exact = False
        floor = max(float(np.abs(npz[f"{n}/{f}"] - npz[f"{n}/fp32/{f}"]).max()) for n in names)
        scale = float(np.abs(ref).max())
        assert float(np.abs(got - ref).max()) <= 0.5 * floor + 4e-3 * scale, (prefix, f)
    return exact


def _threads(meta):
    # oneDNN's bf16 GEMM partitions work by thread count, so the reference's own last bits depend on it:
    # replay the fixture under the thread count it was generated with
    torch.set_num_threads(int(meta.get("cpu_threads", torch.get_num_threads())))


def test_oracle_reproduces_reference_tiny():
    npz, meta = load_golden("tiny")
    _threads(meta)
    names = [c["name"] for c in meta["cases"]]
    n_exact = 0
    for case in meta["cases"]:
        cfg = make_cfg("tiny", case["image_size"], case["vit_image_size"])
        sd = synth.synth_state_dict(cfg, seed=case["weight_seed"], dtype=torch.float32)
        sd = {k: v.to(torch.bfloat16) for k, v in sd.items()}
        px, ids, mask, _ = case_inputs_masked(cfg, case)     # (incl. the left-padded and the holey mask of round 6)
        probes = {}
        out = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID, probes=probes)
        n_exact += _check(out, npz, case["name"], names)
        for k, v in probes.items():
            ref = npz[f"{case['name']}/probe/{k}"]
            assert v.shape == ref.shape
            assert np.abs(v.float().numpy() - ref).max() <= 0.05 * np.abs(ref).max() + 1e-3, (case["name"], k)
        assert out["score"].dtype == torch.float32 and out["aspect_scores"].dtype == torch.float32
        assert out["rewards"].dtype == torch.bfloat16 and out["weighted_scores"].dtype == torch.bfloat16
    print(f"bit-exact cases: {n_exact}/{len(meta['cases'])}")


def test_oracle_padded_batch_equals_single():
    """right-padded batch == per-sample forwards (the property that lets the HIP path pack sequences)"""
    npz, meta = load_golden("tiny")
    case = next(c for c in meta["cases"] if c["name"] == "batch2pad")
    cfg = make_cfg("tiny", case["image_size"])
    sd = {k: v.to(torch.bfloat16) for k, v in synth.synth_state_dict(cfg, seed=case["weight_seed"], dtype=torch.float32).items()}
    px, ids, mask, single_ids = case_inputs(cfg, case["videos"], case["pixel_seed"], case["image_size"])
    both = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    off = 0
    for b, v in enumerate(case["videos"]):
        one = ref_cpu.reward_forward(sd, cfg, px[off:off + v["n_tiles"]], single_ids[b], None, synth.IMG_CONTEXT_ID, synth.PAD_ID)
        off += v["n_tiles"]
        for f in FIELDS:
            assert torch.equal(one[f][0], both[f][b]), (f, b)


def test_oracle_full_c1_one_video():
    """MJ-VIDEO-2B dimensions (2.2 B parameters), 8 frames @224: one video against the reference's output"""
    npz, meta = load_golden("full_c1")
    _threads(meta)
    cfg = make_cfg("2b", 224)
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"])
    v = meta["videos"][0]
    px, ids, mask, _ = case_inputs(cfg, [v], meta["pixel_seed"], 224)
    out = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    _check(out, npz, "v0", ["v0"])


def test_oracle_error_cases():
    cfg = make_cfg("tiny", 56)
    sd = {k: v.to(torch.bfloat16) for k, v in synth.synth_state_dict(cfg, seed=3, dtype=torch.float32).items()}
    px = synth.synth_pixel_values(1, 0, 2, 56)
    ids = synth.synth_input_ids(8, 1)
    bad = ids.clone()
    bad[0, -1] = 5
    with pytest.raises(ValueError, match="Token pattern not found"):
        ref_cpu.reward_forward(sd, cfg, px, bad, None, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    with pytest.raises(ValueError, match="Cannot handle batch sizes > 1"):
        ref_cpu.reward_forward(sd, cfg, torch.cat([px, px]), torch.cat([ids, ids]), None, synth.IMG_CONTEXT_ID, None)


def _ntk_cfg(meta):
    from mj_video_amd import configuration as C
    cd = C.tiny_config_dict(meta["image_size"])
    cd["llm_config"]["max_position_embeddings"] = meta["max_position_embeddings"]
    cd["llm_config"]["rope_scaling"] = dict(meta["rope_scaling"])
    return C.InternVLChatRewardModelingConfig(**cd, **C.mjvideo_head_kwargs())


def test_oracle_reproduces_the_sticky_dynamic_ntk_sequence():
    """tests/golden/layers.npz "ntk/*": ONE reference model (max_position_embeddings 48, rope_scaling dynamic) scored
    short -> long -> short -> a padded batch; its rotary cache regrows on longer inputs and the rescaled inv_freq sticks
    (modeling_internlm2.py:169-176,204-229), so the third call differs from the first.  The oracle with a rope_state kept
    across the calls reproduces all four; a fresh state per call reproduces only the calls that regrow the cache."""
    npz, meta = load_golden("layers")
    _threads(meta)
    m = meta["ntk"]
    cfg = _ntk_cfg(m)
    sd = synth.synth_state_dict(cfg, seed=m["weight_seed"])
    names = [c["name"] for c in m["calls"]]
    state = {}
    for call in m["calls"]:
        px, ids, mask, _ = case_inputs(cfg, call["videos"], m["pixel_seed"], m["image_size"])
        assert int(ids.shape[1]) == call["n_tokens"]
        out = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID, rope_state=state)
        _check(out, npz, f"ntk/{call['name']}", [f"ntk/{n}" for n in names])
    assert not np.array_equal(npz["ntk/short_first/hidden_state"], npz["ntk/short_again/hidden_state"])
    # stateless replay of the third call = the FIRST call's numbers, not the third's
    call = m["calls"][2]
    px, ids, mask, _ = case_inputs(cfg, call["videos"], m["pixel_seed"], m["image_size"])
    fresh = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    assert np.array_equal(fresh["hidden_state"].float().numpy(), npz["ntk/short_first/hidden_state"])


# ------------------------------------------------------------------ MXFP8 operand rounding (oracle/ref_fp8.py)
def test_fp8_oracle_element_encoder_equals_torch_cast_and_fast_path():
    """oracle/ref_fp8.py is a DEFINITION (the reference has no fp8 path): its table-driven e4m3 encoder is held to torch's
    float8_e4m3fn cast on every bf16 value inside the format's range, its block quantiser to the properties the format
    promises (no saturation, error <= half an e4m3 ulp of the block maximum's binade, power-of-two scales, zero blocks),
    and the multi-threaded form used at 2B dims to the table form bit for bit."""
    import numpy as np
    import torch
    from oracle import ref_fp8 as R
    u = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()
    m = torch.isfinite(u) & (u.abs() <= 448)
    mine = R.e4m3_encode(u[m].numpy())
    tc = u[m].to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(mine, tc)
    assert np.array_equal(R.e4m3_encode(R.e4m3_decode(np.arange(127, dtype=np.uint8))), np.arange(127, dtype=np.uint8))
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(96, 512, generator=g) * torch.logspace(-12, 12, 96, base=2.0)[:, None]).to(torch.bfloat16)
    x[3, :32] = 0.0
    x[5, 64:96] = torch.tensor([1.75] + [0.2] * 31)
    x[6, 64:96] = torch.tensor([1.7578125] + [0.2] * 31)
    q, sb = R.mx_quantize(x)
    assert ((q & 0x7F) <= 0x7E).all() and sb.min() >= 1
    d = R.mx_dequantize(q, sb)
    blk_amax = x.float().abs().reshape(96, 16, 32).amax(2)
    err = (d - x.float()).abs().reshape(96, 16, 32).amax(2)
    assert (err <= blk_amax * (2.0 ** -4) + 1e-30).all()                   # half an ulp of a 3-bit mantissa, at worst 2^-4 of amax
    assert sb[5, 2] + 1 == sb[6, 2]                                        # 1.75 * 2^e fits, the next bf16 above moves the scale up
    assert np.array_equal(sb[3, :1], np.array([1], dtype=np.uint8)) and (q[3, :32] == 0).all()
    assert torch.equal(R.mx_fake_quant(x), R.mx_fake_quant_fast(x))
    rec = R.mx_scale_records(sb)
    assert rec.size == (512 // 128) * 2 * 256 and rec[(0 * 2 + 0) * 256 + 5 * 16 + 2 * 4 + 0] == sb[5, 2]
    assert rec[(1 * 2 + 1) * 256 + (70 % 16) * 16 + 1 * 4 + ((70 >> 4) & 3)] == sb[70, 5]


def test_fp8_oracle_hook_leaves_the_bf16_oracle_untouched():
    """the fp8 mode swaps ONE name in ref_cpu (the five FFN Linears) and restores it: outside the context the pinned bf16
    oracle computes exactly what it did (tiny fixture, bit for bit), inside it something else"""
    import torch
    from mj_video_amd import synth
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    from oracle import ref_cpu, ref_fp8
    from util import make_cfg
    cfg = make_cfg("tiny", 56)
    sd = synth.synth_state_dict(cfg, seed=3)
    px = synth.synth_pixel_values(4, 0, 2, 56)
    ids = synth.synth_input_ids(2 * num_image_tokens_per_tile(cfg), 1)
    a = ref_cpu.reward_forward(sd, cfg, px, ids, torch.ones_like(ids), synth.IMG_CONTEXT_ID, synth.PAD_ID)
    b = ref_fp8.reward_forward_fp8(sd, cfg, px, ids, torch.ones_like(ids), synth.IMG_CONTEXT_ID, synth.PAD_ID)
    b2 = ref_fp8.reward_forward_fp8(sd, cfg, px, ids, torch.ones_like(ids), synth.IMG_CONTEXT_ID, synth.PAD_ID, fast=False)
    c = ref_cpu.reward_forward(sd, cfg, px, ids, torch.ones_like(ids), synth.IMG_CONTEXT_ID, synth.PAD_ID)
    assert ref_cpu._ffn_linear is torch.nn.functional.linear
    for k in a:
        assert torch.equal(a[k], c[k]), k
        assert torch.equal(b[k], b2[k]), k
    assert not torch.equal(a["hidden_state"], b["hidden_state"])
