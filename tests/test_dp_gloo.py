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


def _bench_worker(rank, world, port, pairs_per_rank, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    global_pairs = list(range(world * pairs_per_rank))
    seen = []

    def score_local(local):          # the model's place: this rank's shard only
        seen.append(list(local))
        return fake_scores([1000 + p for p in local])

    out = bench.dp_step(score_local, global_pairs, torch.device("cpu"))
    q.put((rank, out, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_function_world2():
    """bench.py's step at N > 1 (dp_step -> parallel.score_pairs_dp: contiguous shards, one all-gather) at world size 2 over
    gloo: every rank ends with the whole [pairs, 2, 34] block in pair order, each rank scored exactly its own 8 pairs
    (BASELINE configs[2]'s shard size)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = fake_scores([1000 + p for p in range(16)])
    for rank, out, seen in results:
        assert torch.equal(out, expect), rank
        assert seen == [list(range(8 * rank, 8 * rank + 8))]


def _bad_width_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def score_fn(local):             # rank 1 misbehaves; with 1 pair rank 1's shard is EMPTY, with 3 it is not
        return torch.zeros(len(local), 2, 34 if rank == 0 else 33)

    outcome = []
    for n in (3, 1):
        try:
            parallel.score_pairs_dp(score_fn, list(range(n)), device=torch.device("cpu"))
            outcome.append("ok")
        except ValueError as e:
            outcome.append("ValueError")
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_wrong_width_raises_on_every_rank_after_the_collective():
    """a rank whose score_fn returns the wrong width must not raise BEFORE the all-gather (the other ranks - an empty-shard
    rank included - would stay blocked in it): every rank raises after the collective; a rank with an empty shard never
    calls score_fn, so a 1-pair batch passes"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bad_width_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == ["ValueError", "ok"] and results[1] == ["ValueError", "ok"]


def _raising_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def score_fn(local):             # rank 1's model throws (a bad input, an out-of-memory ...); rank 0 is fine
        if rank == 1:
            raise RuntimeError("boom on rank 1")
        return fake_scores(local)

    try:
        parallel.score_pairs_dp(score_fn, list(range(4)), device=torch.device("cpu"))
        outcome = "ok"
    except RuntimeError as e:
        outcome = f"RuntimeError: {e}"
    except ValueError as e:
        outcome = "ValueError" + (" names rank 1" if "[1]" in str(e) else "")
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_exception_in_score_fn_reaches_every_rank_after_the_collective():
    """an exception inside one rank's score_fn must not leave the other ranks blocked in the all-gather: the failing rank
    still joins the collective (status row = 2) and re-raises its own exception afterwards, the others raise a ValueError that
    names it"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_raising_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] == "RuntimeError: boom on rank 1" and results[0] == "ValueError names rank 1", results


def _world8_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = {}
    for n_pairs in (64, 61, 3):
        pairs = list(range(500, 500 + n_pairs))
        calls = []

        def score_fn(local):
            calls.append(list(local))
            return fake_scores(local)

        out = parallel.score_pairs_dp(score_fn, pairs, device=torch.device("cpu"))
        rec[n_pairs] = (out, calls)
    # error path at world 8: rank 5's model throws on a 61-pair batch; the others must come back with a ValueError naming it
    def bad(local):
        if rank == 5:
            raise RuntimeError("boom on rank 5")
        return fake_scores(local)
    try:
        parallel.score_pairs_dp(bad, list(range(61)), device=torch.device("cpu"))
        rec["error"] = "ok"
    except RuntimeError as e:
        rec["error"] = f"RuntimeError: {e}"
    except ValueError as e:
        rec["error"] = "ValueError" + (" names rank 5" if "[5]" in str(e) else str(e))
    # ... and the group still works afterwards (nobody is left inside a collective)
    rec["after"] = parallel.score_pairs_dp(fake_scores, list(range(16)), device=torch.device("cpu"))
    q.put((rank, rec))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world8_dry_run():
    """The first hardware SCALE run is 8 ranks (BASELINE configs[2]: 64 pairs over 8 MI355X) and has never executed on GPUs:
    this is its sharding arithmetic at world size 8 over gloo - 64 pairs (8 per rank, bench.py's shard), 61 (uneven: five ranks
    take 8, three take 7) and 3 (five ranks with EMPTY shards): every rank ends with the whole block in pair order bit for bit,
    every pair is scored exactly once and by the rank that owns its contiguous block, a throwing rank reaches everyone after the
    collective, and the group keeps working."""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for n_pairs in (64, 61, 3):
        pairs = list(range(500, 500 + n_pairs))
        single = fake_scores(pairs)
        seen = []
        for rank in range(world):
            out, calls = results[rank][n_pairs]
            assert torch.equal(out, single), (n_pairs, rank)
            lo, hi = parallel.shard_bounds(n_pairs, world, rank)
            assert calls == ([pairs[lo:hi]] if hi > lo else []), (n_pairs, rank, calls)
            seen += [x for c in calls for x in c]
        assert seen == pairs                                   # rank order == pair order, each pair exactly once
    sizes = [parallel.shard_bounds(61, world, r) for r in range(world)]
    assert [b - a for a, b in sizes] == [8, 8, 8, 8, 8, 7, 7, 7] and sizes[0][0] == 0 and sizes[-1][1] == 61
    assert [b - a for a, b in (parallel.shard_bounds(3, world, r) for r in range(world))] == [1, 1, 1, 0, 0, 0, 0, 0]
    for rank in range(world):
        assert results[rank]["error"] == ("RuntimeError: boom on rank 5" if rank == 5 else "ValueError names rank 5"), results[rank]["error"]
        assert torch.equal(results[rank]["after"], fake_scores(list(range(16))))
