"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py (shard ranges, barrier, max-over-ranks timing,
optional global min/max) -- the data path itself has no collective (replicas only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paif_amd import dist_utils as D, synthetic as S

    r, lr, w = D.env_world()
    start, n = D.shard_range(r, w, 2)
    ir, vis, _ = S.make_batch(n, 16, 24, start=start)
    dist.barrier()
    dt = D.max_over_ranks(0.5 + 0.25 * rank, dist)            # slowest rank wins
    mn, mx = D.global_minmax(torch.tensor(float(ir.min())), torch.tensor(float(ir.max())), dist)
    q.put((rank, start, n, dt, float(mn), float(mx), float(ir.min()), float(ir.max()), float(ir.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_plumbing_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, n0, dt0, mn0, mx0, lmn0, lmx0, sum0), (r1, s1, n1, dt1, mn1, mx1, lmn1, lmx1, sum1) = res
    assert (s0, n0, s1, n1) == (0, 2, 2, 2)                   # disjoint shards, fixed per-rank batch (weak scaling)
    assert sum0 != sum1                                        # the two ranks really hold different samples
    assert dt0 == dt1 == 0.75                                  # MAX over ranks
    assert mn0 == mn1 == min(lmn0, lmn1) and mx0 == mx1 == max(lmx0, lmx1)


def test_single_process_is_identity():
    from paif_amd import dist_utils as D

    assert D.max_over_ranks(1.25) == 1.25
    a, b = D.global_minmax(torch.tensor(0.1), torch.tensor(0.9))
    assert float(a) == pytest.approx(0.1) and float(b) == pytest.approx(0.9)
