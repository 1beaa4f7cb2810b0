"""CPU, world_size 2 over gloo: the data-parallel sharding + all-gather of mj_video_amd.parallel reproduces the
single-process result bit for bit and in the original pair order (the N>1 path of bench.py / SURVEY.md §8(e))."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mj_video_amd import parallel


def fake_scores(pairs):
    """stand-in for the model: a deterministic [n, 2, 34] block that depends only on each pair's id"""
    out = torch.zeros(len(pairs), 2, 34)
    for i, p in enumerate(pairs):
        g = torch.Generator().manual_seed(int(p))
        out[i] = torch.randn(2, 34, generator=g)
    return out


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = list(range(100, 100 + n_pairs))
    calls = []

    def score_fn(local):
        calls.append(list(local))
        return fake_scores(local)

    out = parallel.score_pairs_dp(score_fn, pairs, device=torch.device("cpu"))
    q.put((rank, out, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 5, 1])
def test_dp_two_ranks_equals_single_process(n_pairs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = parallel.score_pairs_dp(fake_scores, list(range(100, 100 + n_pairs)))
    seen = []
    for rank, out, calls in results:
        assert torch.equal(out, single), rank          # every rank holds the full block, original order, bitwise
        seen += [x for c in calls for x in c]
    assert sorted(seen) == list(range(100, 100 + n_pairs))  # each pair scored exactly once across ranks
