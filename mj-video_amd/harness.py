"""Pairwise-preference evaluation protocol of the reference's only reward-model driver.

Follows scripts/eval/eval_genai_mjvideo.py:118-165: the same caption scored against the left and the right
video (two independent forwards), right wins iff ``score_right > score_left``, a video is "good" iff its score
is > 0, a tie vote is correct iff both are good, a both-bad vote iff both are bad; ``prefer_Acc`` counts only
left/right votes, ``Acc`` all four vote types.  The scoring itself is batched (and data-parallel through
mj_video_amd.parallel) instead of the reference's batch-1 Python loop.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import torch

from .chat_input import prepare_chat_input, video_prefix

VOTE_TYPES = ("leftvote", "rightvote", "bothbad_vote", "tievote")


@dataclass
class PreferenceCounts:
    prefer_truth: int = 0
    prefer_total: int = 0
    truth: int = 0
    total: int = 0

    def update(self, vote_type: str, score_left: float, score_right: float) -> None:
        left_good, right_good = score_left > 0, score_right > 0
        if vote_type == "rightvote":
            self.prefer_total += 1
            self.total += 1
            if score_right > score_left:
                self.prefer_truth += 1
                self.truth += 1
        elif vote_type == "leftvote":
            self.prefer_total += 1
            self.total += 1
            if score_right < score_left:
                self.prefer_truth += 1
                self.truth += 1
        elif vote_type == "bothbad_vote":
            self.total += 1
            if not left_good and not right_good:
                self.truth += 1
        elif vote_type == "tievote":
            self.total += 1
            if left_good and right_good:
                self.truth += 1

    @property
    def prefer_acc(self) -> float:
        return self.prefer_truth / self.prefer_total if self.prefer_total else float("nan")

    @property
    def acc(self) -> float:
        return self.truth / self.total if self.total else float("nan")


def evaluate_votes(votes: Iterable[Tuple[str, float, float]]) -> PreferenceCounts:
    c = PreferenceCounts()
    for vote_type, sl, sr in votes:
        c.update(vote_type, float(sl), float(sr))
    return c


def pad_right(ids_list: Sequence[torch.Tensor], pad_id: int):
    n = max(int(t.shape[-1]) for t in ids_list)
    ids = torch.full((len(ids_list), n), pad_id, dtype=torch.long)
    mask = torch.zeros((len(ids_list), n), dtype=torch.long)
    for i, t in enumerate(ids_list):
        k = int(t.shape[-1])
        ids[i, :k] = t.reshape(-1).cpu()
        mask[i, :k] = 1
    return ids, mask


@torch.no_grad()
def score_pair_batch(model, config, tokenizer, examples: Sequence[dict], generation_config: dict) -> torch.Tensor:
    """examples: dicts with ``prompt``, ``left_pixels``, ``right_pixels`` (each [F,3,S,S] bf16).  Both videos of
    every pair go through ONE packed forward; returns ``[len(examples), 2, 1 + n_aspects + n_objectives]`` fp32."""
    dev = model.model.device
    px, ids = [], []
    for ex in examples:
        for key in ("left_pixels", "right_pixels"):
            pv = ex[key].to(torch.bfloat16)
            q = video_prefix(pv.shape[0]) + ex["prompt"]
            i, _ = prepare_chat_input(config, tokenizer, pv, q, generation_config, device="cpu")
            px.append(pv)
            ids.append(i)
    pad_id = model.config.pad_token_id
    if pad_id is None:
        raise ValueError("model.config.pad_token_id must be set to batch pairs (eval_genai_mjvideo.py:112)")
    ids_b, mask = pad_right(ids, pad_id)
    model.forward(torch.cat(px).to(dev), ids_b.to(dev), mask.to(dev))
    return model.last_packed34.view(len(examples), 2, -1)


# ---------------------------------------------------------------------------------------------------------------
# Collator batch format and MJ-BENCH-VIDEO label schema
@torch.no_grad()
def score_collated_batch(model, batch: dict):
    """Scores one batch in the training collator's layout (scripts/data_processor/dataset.py:445-554 as consumed by
    scripts/train/overall_train.py:70-82): ``video_{0,1}_pixel_values [B, F, 3, H, W]``, right-padded
    ``video_{0,1}_input_ids [B, N]`` and ``video_{0,1}_attention_mask``.  The 5-D pixel block is flattened to
    ``[B*F, 3, H, W]`` exactly as the trainer does; both videos of every pair go through one packed forward.
    Returns ``(output_video_0, output_video_1)`` (CustomOutput each)."""
    dev = model.model.device
    outs = []
    px = [batch[f"video_{i}_pixel_values"] for i in (0, 1)]
    b, f, c, h, w = px[0].shape
    flat = torch.cat([p.reshape(-1, c, h, w) for p in px]).to(torch.bfloat16).to(dev)
    ids = [batch[f"video_{i}_input_ids"] for i in (0, 1)]
    n = max(int(t.shape[1]) for t in ids)
    pad = model.config.pad_token_id
    if pad is None:
        raise ValueError("model.config.pad_token_id must be set for collated batches (dataset.py pads with it)")

    def widen(t, fill):
        if t.shape[1] == n:
            return t
        extra = torch.full((t.shape[0], n - t.shape[1]), fill, dtype=t.dtype, device=t.device)
        return torch.cat([t, extra], dim=1)

    ids_all = torch.cat([widen(t, pad) for t in ids])
    mask_all = torch.cat([widen(batch[f"video_{i}_attention_mask"], 0) for i in (0, 1)])
    out = model.forward(flat, ids_all.to(dev), mask_all.to(dev))
    from .modeling import CustomOutput
    from dataclasses import fields
    for i in (0, 1):
        sl = slice(i * b, (i + 1) * b)
        outs.append(CustomOutput(**{fl.name: getattr(out, fl.name)[sl] for fl in fields(CustomOutput)}))
    return outs[0], outs[1]


def criteria_targets(labels: dict, mse: bool = True):
    """MJ-BENCH-VIDEO per-criterion labels -> (score, relevance, names): 1 = good -> (+1, relevant), 2 = bad ->
    (-1 with ``mse`` else 0, relevant), anything else -> (0, irrelevant)   (dataset.py:52-85)."""
    score, related, names = [], [], []
    for key, value in labels.items():
        names.append(key)
        score.append(1 if value == 1 else ((-1 if mse else 0) if value == 2 else 0))
        related.append(1 if value in (1, 2) else 0)
    return score, related, names


def overall_target(value, mse: bool = True):
    """Same mapping for a single overall label (dataset.py:57-69)."""
    return ([1], [1]) if value == 1 else (([-1 if mse else 0], [1]) if value == 2 else ([0], [0]))


def preference_targets(labels):
    """"Video 1 better" -> (0, counted), "Video 2 better" -> (1, counted), anything else -> (1, masked out)
    (dataset.py:87-112); ``labels`` is a dict of per-aspect strings or a single overall string."""
    values = list(labels.values()) if isinstance(labels, dict) else [labels]
    pref = [0 if v == "Video 1 better" else 1 for v in values]
    mask = [1 if v in ("Video 1 better", "Video 2 better") else 0 for v in values]
    return pref, mask
