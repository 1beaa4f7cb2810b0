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


def score_pairs_dp(score_fn: Callable[[Sequence], torch.Tensor], pairs: Sequence, group: Optional[dist.ProcessGroup] = None,
                   device: Optional[torch.device] = None) -> torch.Tensor:
    """Scores ``pairs`` data-parallel and returns the full ``[len(pairs), 2, W]`` fp32 block on every rank, in the
    original pair order (bitwise what a single rank would produce, since per-video math is unchanged).

    ``score_fn(local_pairs) -> [len(local_pairs), 2, W]`` runs the model on this rank's shard (W = 34 for MJ-VIDEO).
    Works without an initialised process group (single process)."""
    if not (dist.is_available() and dist.is_initialized()):
        return score_fn(pairs).float()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(len(pairs), world, rank)
    local = score_fn(pairs[lo:hi]).float() if hi > lo else None
    per = -(-len(pairs) // world)  # all_gather_into_tensor needs equal blocks: pad to the largest shard
    if local is None:
        if device is None:
            raise ValueError("a rank with an empty shard needs `device` to build its (padding) block")
        width = None
    else:
        device = local.device
        width = local.shape[-1]
    w = torch.tensor([width or 0], dtype=torch.int64, device=device)
    dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
    width = int(w.item())
    block = torch.zeros(per, 2, width, dtype=torch.float32, device=device)
    if local is not None:
        block[:hi - lo] = local
    out = torch.empty(world * per, 2, width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, block, group=group)
    rows = []
    for r in range(world):
        a, b = shard_bounds(len(pairs), world, r)
        rows.append(out[r * per:r * per + (b - a)])
    return torch.cat(rows, dim=0)
