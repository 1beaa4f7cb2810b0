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
