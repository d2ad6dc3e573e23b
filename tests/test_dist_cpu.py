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


# ---------------------------------------------------------------------------------------------
# configs[4]: bucketed gradient all-reduce over the flat gradient arena (world_size 2, gloo)
# ---------------------------------------------------------------------------------------------
def _tiny_model(seed=0):
    g = torch.Generator().manual_seed(seed)
    layers = torch.nn.ModuleList([torch.nn.Linear(40, 600), torch.nn.Linear(600, 700), torch.nn.Linear(700, 30)])   # 24k / 420k / 21k weights
    for l in layers:
        with torch.no_grad():
            l.weight.copy_(torch.randn(l.weight.shape, generator=g) * 0.05)
            l.bias.copy_(torch.randn(l.bias.shape, generator=g) * 0.05)
    dead = torch.nn.Parameter(torch.ones(5000))          # never receives a gradient (cf. WeTr.classifier.weight)
    dead._paif_never_grad = True
    return layers, dead


def _tiny_loss(layers, x, y):
    h = x
    for i, l in enumerate(layers):
        h = l(h)
        if i < 2:
            h = torch.tanh(h)
    return ((h - y) ** 2).mean()


def _tiny_data(n):
    g = torch.Generator().manual_seed(99)
    return torch.randn(n, 40, generator=g), torch.randn(n, 30, generator=g)


def _reduce_worker(rank, world, port, q, order):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paif_amd import ops
    from paif_amd.dist_utils import GradAllReduce
    from paif_amd.utils.optimizer import ParamArena, arena_order

    layers, dead = _tiny_model()
    x, y = _tiny_data(8)
    per = 8 // world
    xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
    params = [p for l in layers for p in l.parameters()] + [dead]
    arena = ParamArena(arena_order([(p, 0) for p in params]), with_moments=False)
    # arena order = reversed registration order (the last layer's gradients are final first); the dead parameter in the tail
    assert arena.entries[0][0] is layers[2].bias and arena.entries[-1][0] is dead
    red = GradAllReduce(arena, milestones=list(layers), bucket_mb=0.05)      # 13k-float buckets -> one bucket per layer here
    assert len(red.buckets) >= 3 and red.buckets[-1][1] == red.live_end == arena.range_of[id(layers[0].weight)][1]
    early = []
    for it in range(2):                                                      # two backward passes: reset() works
        arena.zero_grad()
        # a stale milestone outside begin()/finish() (what an attack's dgrad-only reverse pass would have produced before the
        # call sites were guarded): must neither launch a bucket nor change what the real backward sends
        red.mark_ready(layers[0]); red.mark_ready(layers[2])
        assert red.launched == [] and red.next_bucket == 0
        red.begin()
        grads = torch.autograd.grad(_tiny_loss(layers, xs, ys), [p for l in layers for p in l.parameters()])
        gmap = dict(zip([id(p) for l in layers for p in l.parameters()], grads))
        for li in order:                                                     # the reverse pass reaches the milestones in this order
            for p in layers[li].parameters():
                ops.grad_of(p).copy_(gmap[id(p)])                            # what a wgrad kernel does: write the arena slot
            red.mark_ready(layers[li])
            early.append(len(red.launched))
        red.finish()
    # numpy arrays: pickled by value.  torch tensors travel through a multiprocessing queue as file descriptors served by the SENDING
    # process, and the parent may fetch them after this worker has exited (FileNotFoundError on the worker's listener socket)
    out = {i: [p.grad.detach().clone().numpy() for p in layers[i].parameters()] for i in range(3)}
    q.put((rank, out, early, float(arena.grad[arena.range_of[id(dead)][0]:].abs().sum()), dead.grad is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("order", [(2, 1, 0), (0, 2, 1)])
def test_bucketed_gradient_allreduce_gloo(order):
    """Every gradient element is reduced exactly once and averaged: the result equals the single-process gradient of the
    whole batch, whatever order the milestones arrive in; buckets whose parameters are final are launched before the
    backward ends; the never-grad tail is not communicated."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, q, order)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    layers, _ = _tiny_model()
    x, y = _tiny_data(8)
    ref = torch.autograd.grad(_tiny_loss(layers, x, y), [p for l in layers for p in l.parameters()])
    refs = {0: ref[0:2], 1: ref[2:4], 2: ref[4:6]}
    for rank, out, early, dead_sum, dead_none in res:
        for i in range(3):
            for a, b in zip(out[i], refs[i]):
                assert float((torch.from_numpy(a) - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
        assert dead_sum == 0.0 and dead_none
        if order == (2, 1, 0):       # backward order: buckets go out while later layers are still "computing"
            assert early[0] >= 1 and early[1] > early[0]
        else:                        # layer 0 first: nothing is contiguous-ready until layer 2 arrives
            assert early[0] == 0
    assert res[0][2] == res[1][2]    # both ranks launched the same buckets at the same points (no deadlock by construction)


def test_bench_self_launch_refuses_without_enough_gpus():
    """`python bench.py --gpus N` outside a torch.distributed.run environment starts the N ranks itself (bench.self_launch: child
    processes, the parent never touches the GPU).  On a host that exposes fewer than N GPUs it must say so and exit 2 -- before
    spawning anything."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "GPU(s)" in r.stderr and '"metric"' not in r.stdout


def test_bench_self_launch_refuses_under_a_profiler_preload():
    """under rocprofv3 the preloaded tool library has initialised the GPU before bench.py starts; spawning the ranks from
    there would be a launcher hop from a GPU-initialised process.  bench.self_launch refuses (exit 2) before spawning anything."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "profiler" in r.stderr and '"metric"' not in r.stdout


def test_bench_also_child_waits_for_its_parent_and_leaves_quietly():
    """bench.py measures the `also` block (configs[2]-[4]) in a child process that the headline process starts BEFORE it touches the GPU
    (a fault there must not lose the headline).  The child imports torch, then blocks on its stdin: end-of-file (the parent
    died, or never asked) makes it exit 0 without a word and without initialising a GPU."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--also-child"], env=env, input="", capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "", (r.returncode, r.stdout[-300:], r.stderr[-300:])
    # asked to go on a box without a GPU: one JSON object with the reason, exit 1 -- the parent turns that into also = {"error": ...}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--also-child"], env=env, input="go\n", capture_output=True, text=True,
                       timeout=300)
    import json
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 1 and "error" in json.loads(r.stdout.strip().splitlines()[-1])


def test_bootstrap_interval_of_the_argmax_clause_is_seeded_and_ordered():
    """tests/test_f16_storage_gpu.py asserts SURVEY 8(d)'s 99.9 % argmax clause on the LOWER end of a percentile bootstrap over samples:
    the interval is reproducible (seeded), ordered, contains the aggregate, and widens with the sample-to-sample spread."""
    import importlib
    m = importlib.import_module("tests.test_f16_storage_gpu")
    moved = [280, 342, 104, 85, 169, 541, 100, 125, 64, 45, 289, 148, 147, 66, 277, 134, 196, 205, 155, 68, 67, 100, 376, 143, 124, 111, 292, 154,
             189, 167, 105, 366]           # the fp16 configuration on MI355X, round 6 (profiles/r06_f16_storage_report.json)
    lo, med, hi = m._bootstrap_interval(moved)
    assert (lo, med, hi) == tuple(m._bootstrap_interval(moved))
    agg = 1.0 - sum(moved) / (32 * 307200.0)
    assert lo < agg < hi and lo <= med <= hi and abs(med - agg) < 2e-5
    assert 0.9992 < lo < 0.9994 and 0.9995 < hi < 0.9996
    lo2, _, hi2 = m._bootstrap_interval([v * 2 if i % 2 else v // 2 for i, v in enumerate(moved)])
    assert hi2 - lo2 > hi - lo
