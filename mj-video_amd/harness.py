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

import numpy as np
import torch

from . import _lib
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


def prefetch_to_device(batches: Iterable, device, depth: int = 1):
    """Yields the items of ``batches`` (tensors, or tuples / lists / dicts of tensors; anything else passes through) moved to
    ``device``, with the upload of the NEXT ``depth`` items already running on a copy stream of its own: batch i + 1 crosses
    PCIe while batch i is scored, so a caller that starts from host memory sees the HBM-resident rate (without this the copies
    sit on the scoring stream: -1.8 % with bf16 pixel tensors, -4.5 % with uint8 720p frames, profiles/r03_f_pcie_inclusive.txt).
    Host tensors are pinned here if they are not yet (an unpinned source makes the copy synchronous).  The yielded tensors are
    safe to use on the current stream (it waits for the copy's event; ``record_stream`` keeps the allocator from recycling
    them under the scoring kernels)."""
    from collections import deque
    device = torch.device(device)
    side = torch.cuda.Stream(device=device)

    def move(x, fn):
        if isinstance(x, torch.Tensor):
            return fn(x)
        if isinstance(x, dict):
            return {k: move(v, fn) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(move(v, fn) for v in x)
        return x

    def start(item):
        def up(t):
            if t.is_cuda:
                return t
            return (t if t.is_pinned() else t.pin_memory()).to(device, non_blocking=True)
        with torch.cuda.stream(side):
            dev_item = move(item, up)
            ev = torch.cuda.Event()
            ev.record(side)
        return dev_item, ev

    def finish(entry):
        dev_item, ev = entry
        cur = torch.cuda.current_stream(device)
        cur.wait_event(ev)
        move(dev_item, lambda t: (t.record_stream(cur), t)[1] if t.is_cuda else t)
        return dev_item

    pending = deque()
    for item in batches:
        pending.append(start(item))
        if len(pending) > depth:
            yield finish(pending.popleft())
    while pending:
        yield finish(pending.popleft())


@torch.no_grad()
def score_pair_batch(model, config, tokenizer, examples: Sequence[dict], generation_config: dict) -> torch.Tensor:
    """examples: dicts with ``prompt``, ``left_pixels``, ``right_pixels`` (each [F,3,S,S] bf16).  Both videos of
    every pair go through ONE packed forward; returns ``[len(examples), 2, 1 + n_aspects + n_objectives]`` fp32."""
    _lib.assert_product_library()   # an evaluation never scores on a diagnostics build of the kernels
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
    _lib.assert_product_library()
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


# ------------------------------------------------------------------------------------ MJ-BENCH-VIDEO (datas/test.json)
ASPECT_KEYS = ("Alignment", "Safety", "Fineness", "Consistency", "Bias")   # order of video_*_overall_score in test.json


@dataclass
class ConfusionCounts:
    """Per-dimension bookkeeping of scripts/train/overall_train.py:308-324 (``calculate_metrics``): predictions are the signs
    of the model's outputs (``> 0``), a label counts as positive when it EQUALS 1 and as negative when it EQUALS 0 (the
    dataset's default labels use -1 for "bad", dataset.py:52-85, which is neither - the reference's own quirk, reproduced),
    and only dimensions with a non-zero relevance mask are counted."""
    dims: int
    correct: np.ndarray = None
    total: np.ndarray = None
    tp: np.ndarray = None
    fp: np.ndarray = None
    tn: np.ndarray = None
    fn: np.ndarray = None

    def __post_init__(self):
        for f in ("correct", "total", "tp", "fp", "tn", "fn"):
            setattr(self, f, np.zeros(self.dims, dtype=np.float64))

    def update(self, predictions, ground_truth, mask) -> None:
        pred = np.asarray(predictions, dtype=bool).ravel()
        gt = np.asarray(ground_truth).ravel()
        active = np.asarray(mask).ravel() != 0
        self.correct += (pred.astype(gt.dtype) == gt) & active
        self.tp += pred & (gt == 1) & active
        self.fp += pred & (gt == 0) & active
        self.tn += ~pred & (gt == 0) & active
        self.fn += ~pred & (gt == 1) & active
        self.total += active

    def summary(self) -> dict:
        """The figures ``save_metrics`` reports (overall_train.py:326-388): pooled accuracy / precision / recall / F1 and the
        per-dimension vectors (a dimension that never occurs divides 0 by 0: NaN, as in the reference's tensors)."""
        def ratio(a, b):
            return float(a) / float(b) if b > 0 else 0
        tp, fp, tn, fn = self.tp, self.fp, self.tn, self.fn
        acc = ratio(self.correct.sum(), self.total.sum())
        rec = ratio(tp.sum(), (tp + fn).sum())
        pre = ratio(tp.sum(), (tp + fp).sum())
        f1 = 2 * (rec * pre) / (rec + pre) if (rec + pre) > 0 else 0
        with np.errstate(divide="ignore", invalid="ignore"):
            acc_d, rec_d, pre_d = self.correct / self.total, tp / (tp + fn), tp / (tp + fp)
            f1_d = 2 * (rec_d * pre_d) / (rec_d + pre_d)
        return dict(accuracy=acc, precision=pre, recall=rec, f1=f1, accuracy_dim=acc_d.tolist(), precision_dim=pre_d.tolist(),
                    recall_dim=rec_d.tolist(), f1_dim=f1_d.tolist(), tp=tp.tolist(), fp=fp.tolist(), tn=tn.tolist(), fn=fn.tolist())


def mjbench_targets(item: dict, mse: bool = True) -> dict:
    """Label tensors of one datas/test.json pair exactly as VideoDataset builds them (dataset.py:329-398): per video the
    criteria / aspect scores and relevance masks, plus the overall preference (0 = video 1 better, 1 = video 2 better) and
    its mask.  The two videos must carry the same label names (dataset.py:346,352)."""
    out = {}
    names = None
    for v in (0, 1):
        cs, cr, cn = criteria_targets(item[f"video_{v}_label"], mse=mse)
        a_s, a_r, an = criteria_targets(item[f"video_{v}_overall_score"], mse=mse)
        if names is not None and names != (cn, an):
            raise ValueError("the two videos of a pair carry different label names")
        names = (cn, an)
        out[f"video_{v}"] = dict(criteria_score=cs, criteria_related=cr, aspect_score=a_s, aspect_related=a_r)
    pref, mask = preference_targets(item["overall_preference"])
    out["overall_preference"], out["overall_mask"] = pref[0], mask[0]
    out["criteria_names"], out["aspect_names"] = names
    return out


def evaluate_mjbench(items: Sequence[dict], scores: "np.ndarray", mse: bool = True) -> dict:
    """MJ-BENCH-VIDEO evaluation of ``scores`` [pairs, 2, 1 + n_aspects + n_criteria] (the packed block of
    ``CustomOutput``: score, aspect_scores, rewards per video - what ``parallel.score_pairs_dp`` gathers) against the
    labels of ``items``: the overall preference accuracy of ``CustomTrainer.evaluate`` (overall_train.py:390-442:
    prefer_predict = not (score_0 > score_1), counted where the pair has a decisive overall preference) and the aspect /
    criteria sign metrics of ``evaluate_aspect`` (overall_train.py:204-306)."""
    scores = np.asarray(scores, dtype=np.float64)
    n_asp = len(items[0]["video_0_overall_score"])
    n_crit = len(items[0]["video_0_label"])
    assert scores.shape == (len(items), 2, 1 + n_asp + n_crit), scores.shape
    aspect, criteria = ConfusionCounts(n_asp), ConfusionCounts(n_crit)
    correct = count = 0
    for i, item in enumerate(items):
        t = mjbench_targets(item, mse=mse)
        for v in (0, 1):
            tv = t[f"video_{v}"]
            aspect.update(scores[i, v, 1:1 + n_asp] > 0, tv["aspect_score"], tv["aspect_related"])
            criteria.update(scores[i, v, 1 + n_asp:] > 0, tv["criteria_score"], tv["criteria_related"])
        prefer_predict = int(not (scores[i, 0, 0] > scores[i, 1, 0]))
        correct += int(prefer_predict == t["overall_preference"]) * t["overall_mask"]
        count += t["overall_mask"]
    return dict(overall_accuracy=(correct / count if count > 0 else 0), overall_correct=correct, overall_count=count,
                aspect=aspect.summary(), criteria=criteria.summary(), aspect_names=list(items[0]["video_0_overall_score"]),
                criteria_names=list(items[0]["video_0_label"]))
