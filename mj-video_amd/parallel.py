"""Data-parallel scoring over the GPUs of one node.

The reference has no multi-GPU inference path (its eval loop is single-process, batch 1:
scripts/eval/eval_genai_mjvideo.py:125-163).  Every (caption, video) forward is independent, so pairs are
sharded in contiguous blocks over ranks (both videos of a pair on the same rank), weights are replicated,
and the ONLY exchange is one all-gather of the ``[pairs_local, 2, 34]`` fp32 block
(score, 5 aspect scores, 28 rewards per video) per batch - RCCL over xGMI on the GPUs (backend "nccl"),
gloo in the CPU tests.  No activation or weight ever crosses a link.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous block [lo, hi) of rank ``rank``; blocks differ in size by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


SCORE_WIDTH = 34   # score, 5 aspect scores, 28 rewards per video


def score_pairs_dp(score_fn: Callable[[Sequence], torch.Tensor], pairs: Sequence, group: Optional[dist.ProcessGroup] = None,
                   device: Optional[torch.device] = None, width: int = SCORE_WIDTH) -> torch.Tensor:
    """Scores ``pairs`` data-parallel and returns the full ``[len(pairs), 2, width]`` fp32 block on every rank, in the
    original pair order.  Per-video math does not depend on the sharding except for fp32 summation order inside the GEMMs
    (tile / split-K choice follows the shard's row count), so a video's numbers on N ranks equal the single-rank ones up
    to that re-association - and bit for bit whenever the shards have the single-rank batch composition.

    ``score_fn(local_pairs) -> [len(local_pairs), 2, width]`` runs the model on this rank's shard.  The block width is a
    property of the model (34 for MJ-VIDEO), known to every rank, so the ONLY collective is one ``all_gather_into_tensor``
    of equal-sized blocks (shards are padded to the largest one, plus one status row per rank so that a rank whose
    ``score_fn`` misbehaves makes EVERY rank raise after the collective instead of leaving the others blocked in it).
    Works without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return score_fn(pairs).float()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(len(pairs), world, rank)
    per = -(-len(pairs) // world)  # all_gather_into_tensor needs equal blocks: pad to the largest shard
    # whatever goes wrong in THIS rank's score_fn - an exception, a block of the wrong shape - is reported AFTER the collective,
    # on every rank: raising before it would leave the other ranks (empty shards included) blocked in the all-gather
    local, bad_width, error = None, None, None
    if hi > lo:
        try:
            local = score_fn(pairs[lo:hi]).float()
        except Exception as e:   # noqa: BLE001 - re-raised below, after the collective
            error = e
    # The block's device is derived the SAME way on every rank, before looking at what score_fn returned: the caller's
    # `device`, else the backend's natural one (nccl: this rank's current GPU; gloo / others: the CPU).  A rank whose score_fn
    # raised has no tensor to take a device from, and a guess that differs from the healthy ranks' would mix devices inside
    # the all-gather - a failure or a hang exactly where the error is supposed to be reported.
    if device is None:
        if dist.get_backend(group) == "nccl" and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        else:
            device = torch.device("cpu")
    device = torch.device(device)
    if local is not None:
        if local.dim() != 3 or local.shape[1] != 2 or local.shape[-1] != width:
            bad_width = tuple(local.shape)
            local = None
        else:
            local = local.to(device)
    # one trailing status row per block: [0, 0] = 0 ok / 1 wrong block shape / 2 score_fn raised
    block = torch.zeros(per + 1, 2, width, dtype=torch.float32, device=device)
    if local is not None:
        block[:hi - lo] = local
    if bad_width is not None or error is not None:
        block[per, 0, 0] = 2.0 if error is not None else 1.0
    out = torch.empty(world * (per + 1), 2, width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, block, group=group)
    out = out.view(world, per + 1, 2, width)
    status = out[:, per, 0, 0].tolist()   # ONE device->host read per step for all ranks' status rows
    if error is not None:
        raise error
    failed = [r for r, c in enumerate(status) if c != 0.0]
    if failed:
        raise ValueError(f"score_fn failed on rank(s) {failed} (status {[status[r] for r in failed]}: 1 = block of the wrong shape, "
                         "2 = exception, raised there)"
                         + (f" (here: {bad_width}, expected [pairs, 2, {width}]; pass width=...)" if bad_width is not None else ""))
    rows = []
    for r in range(world):
        a, b = shard_bounds(len(pairs), world, r)
        rows.append(out[r, :b - a])
    return torch.cat(rows, dim=0)
