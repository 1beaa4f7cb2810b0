#!/usr/bin/env python3
"""Pairwise-preference evaluation of the MJ-VIDEO reward model on the MI355X path.

Counterpart of the reference's only reward-model driver (scripts/eval/eval_genai_mjvideo.py:68-165): same model
set-up order (config kwargs -> construct -> load_state_dict(strict) -> pad_token_id -> bf16 -> img_context_token_id ->
eval), same per-pair protocol (one caption, two videos, F frames each, "Frame{i}: <image>\\n" prefix), same decision rule
and the same two metrics (prefer_Acc, Acc).  Differences: pairs are scored in batches (both videos of every pair in one
packed forward) and sharded data-parallel over the GPUs of a node when launched with torch.distributed.run; frames can be
preprocessed on the GPU; examples come from a local JSON/JSONL file (this build has no network, so the GenAI-Bench hub
dataset of eval_genai_mjvideo.py:118 must be exported first).

Example file: a JSON list (or JSONL) of {"prompt": str, "left_video": path, "right_video": path, "vote_type":
"leftvote" | "rightvote" | "tievote" | "bothbad_vote"}.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mj_video_amd  # noqa: E402,F401
from mj_video_amd import harness, parallel, video  # noqa: E402
from mj_video_amd.configuration import DEFAULT_ASPECT2CRITERIA, InternVLChatRewardModelingConfig  # noqa: E402
from mj_video_amd.modeling import InternVLChatRewardModeling  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_name", required=True, help="local directory with config.json (+ tokenizer files)")
    ap.add_argument("--checkpoint_path", default=None, help="directory holding the MJ-VIDEO *.safetensors checkpoint")
    ap.add_argument("--examples", required=True, help="JSON / JSONL file with the pairs")
    ap.add_argument("--num_segments", type=int, default=8)
    ap.add_argument("--max_num", type=int, default=1)
    ap.add_argument("--pairs_per_batch", type=int, default=4)
    ap.add_argument("--num_objectives", type=int, default=28)
    ap.add_argument("--num_aspects", type=int, default=5)
    ap.add_argument("--gating_temperature", type=float, default=1.0)
    ap.add_argument("--gating_hidden_dim", type=int, default=1024)
    ap.add_argument("--gating_n_hidden", type=int, default=3)
    ap.add_argument("--host_preprocessing", action="store_true", help="PIL preprocessing on the host instead of the GPU kernel")
    return ap.parse_args(argv)


def build_model(args, tokenizer, device):
    """eval_genai_mjvideo.py:73-116, in the same order."""
    config = InternVLChatRewardModelingConfig.from_pretrained(
        args.model_name, num_objectives=args.num_objectives, num_aspects=args.num_aspects,
        aspect2criteria=dict(DEFAULT_ASPECT2CRITERIA), gating_temperature=args.gating_temperature,
        gating_hidden_dim=args.gating_hidden_dim, gating_n_hidden=args.gating_n_hidden)
    # with --checkpoint_path the full state dict is loaded strictly below, so a model dir holding only config + tokenizer is enough
    model = InternVLChatRewardModeling(args.model_name, config, allow_uninitialized=args.checkpoint_path is not None)
    if args.checkpoint_path is not None:
        from safetensors.torch import load_file
        files = sorted(f for f in os.listdir(args.checkpoint_path) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"No safetensors files found in {args.checkpoint_path}")
        model.load_state_dict(load_file(os.path.join(args.checkpoint_path, files[0])), strict=True)
    model.config.pad_token_id = tokenizer.pad_token_id
    model = model.to(torch.bfloat16).to(device)
    model.model.img_context_token_id = tokenizer.convert_tokens_to_ids("<IMG_CONTEXT>")
    model.eval()
    return config, model


def load_host(path, args):
    """what a video is in HOST memory before it crosses PCIe: preprocessed bf16 pixel tiles (--host-preprocessing, the
    reference's load_video) or the decoded uint8 frames (default: resize / normalise run on the GPU)"""
    if args.host_preprocessing:
        pv, _ = video.load_video(path, num_segments=args.num_segments, max_num=args.max_num)
        return pv.to(torch.bfloat16)
    return torch.from_numpy(video.decode_frames(path, num_segments=args.num_segments))


def to_pixels(t, args):
    """device tensor of load_host -> [tiles, 3, S, S] bf16 pixel tiles"""
    if t.dtype == torch.bfloat16:
        return t
    pv, _ = video.load_frames_device(t, max_num=args.max_num)
    return pv


def load_pixels(path, args, device):
    return to_pixels(load_host(path, args).to(device), args)


def evaluate_examples(model, config, tokenizer, examples, pixel_loader, pairs_per_batch=4, generation_config=None,
                      host_loader=None, to_device_pixels=None):
    """Scores ``examples`` (dicts with prompt / left_video / right_video / vote_type), data-parallel when a process group
    exists, and returns (PreferenceCounts, scores [n, 2, W]) - identical on every rank.  With ``host_loader`` (path -> host
    tensor) and ``to_device_pixels`` (device tensor -> pixel tiles) the videos of batch i + 1 are decoded and uploaded on a
    copy stream while batch i is scored (harness.prefetch_to_device); ``pixel_loader`` alone is the serial form."""
    generation_config = generation_config if generation_config is not None else {"max_new_tokens": 1024, "do_sample": True}

    def score_fn(local):
        blocks = []
        starts = range(0, len(local), pairs_per_batch)
        if host_loader is not None:
            host_chunks = ([(host_loader(ex["left_video"]), host_loader(ex["right_video"])) for ex in local[i:i + pairs_per_batch]]
                           for i in starts)
            dev_chunks = harness.prefetch_to_device(host_chunks, model.model.device)
            for i, dev in zip(starts, dev_chunks):
                chunk = [dict(prompt=ex["prompt"], left_pixels=to_device_pixels(l), right_pixels=to_device_pixels(r))
                         for ex, (l, r) in zip(local[i:i + pairs_per_batch], dev)]
                blocks.append(harness.score_pair_batch(model, config, tokenizer, chunk, generation_config).float())
        else:
            for i in starts:
                chunk = [dict(prompt=ex["prompt"], left_pixels=pixel_loader(ex["left_video"]),
                              right_pixels=pixel_loader(ex["right_video"])) for ex in local[i:i + pairs_per_batch]]
                blocks.append(harness.score_pair_batch(model, config, tokenizer, chunk, generation_config).float())
        return torch.cat(blocks) if blocks else torch.zeros(0, 2, parallel.SCORE_WIDTH, device=model.model.device)

    scores = parallel.score_pairs_dp(score_fn, list(examples), device=model.model.device)
    s = scores[..., 0].cpu()
    counts = harness.evaluate_votes((ex["vote_type"], s[i, 0].item(), s[i, 1].item()) for i, ex in enumerate(examples))
    return counts, scores


def main(argv=None):
    args = parse_args(argv)
    from mj_video_amd import _lib
    _lib.assert_product_library()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(args.model_name, trust_remote_code=True)
    config, model = build_model(args, tokenizer, device)
    text = open(args.examples).read().strip()
    examples = json.loads(text) if text.startswith("[") else [json.loads(l) for l in text.splitlines() if l.strip()]
    counts, _ = evaluate_examples(model, config, tokenizer, examples, lambda p: load_pixels(p, args, device), args.pairs_per_batch,
                                  host_loader=lambda p: load_host(p, args), to_device_pixels=lambda t: to_pixels(t, args))
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(f"prefer_Acc: {counts.prefer_acc}")
        print(f"Acc: {counts.acc}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
