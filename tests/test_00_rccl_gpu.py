"""Hardware evidence for the RCCL leg of BASELINE configs[4] on ONE GPU (the 8-GPU scaling run is the driver's).

The file name sorts first on purpose: the child process is started BEFORE this pytest process has touched the GPU (a process
group must come up before any other GPU call, and a GPU-initialised process must never be replaced by another program -- the child
is a plain `subprocess`, the parent stays alive and only reads its output).  If an earlier gpu test already ran in this process the
test skips instead of forking from a GPU-initialised parent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bucketed_allreduce_on_a_one_rank_rccl_group():
    """GradAllReduce's NCCL branch (paif_amd/dist_utils.py `_launch`: event on the compute stream -> side stream ->
    all_reduce(AVG, async) -> work.wait() in finish()) on a real RCCL communicator: the gradients of one adversarial-training
    backward (PGD-2 attack with the hook installed, then `_loss_coupled(...).backward()`) are IDENTICAL with and without the reducer
    (AVG over one rank), every bucket is launched exactly once and in arena order, at least one bucket leaves before `backward()`
    returns, and the attack's input-gradient reverse passes mark no milestone."""
    import torch

    from tests import conftest
    if conftest.GPU_TESTS_STARTED[0] > 1 or torch.cuda.is_initialized():
        msg = "must be the first gpu test of the process: the RCCL child is only started from a parent that has not touched the GPU"
        if os.environ.get("PAIF_REQUIRE_RCCL_TEST") == "1":          # a run that must not lose this evidence silently (-k, xdist, reruns)
            pytest.fail(msg)
        pytest.skip(msg)
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:    # a free port: two runs on one box must not collide
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_child.py")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("NCCL_CHILD ")][-1]
    o = json.loads(line[len("NCCL_CHILD "):])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(o, open(os.path.join(out_dir, "rccl_one_rank.json"), "w"), indent=1)
    assert o["backend"] == "nccl" and o["world"] == 1
    assert o["buckets"] >= 4, o                                       # 1 MB buckets over the 3.7 M-float mit_b0 arena
    assert o["marks_during_attack"] == 0 and o["launched_during_attack"] == 0, o
    assert o["marks_during_backward"] == o["milestones"], o           # every milestone module reported exactly once
    assert o["launched_each_bucket_once"] and o["launch_order_is_arena_order"] and o["launched_total"] == o["buckets"], o
    assert o["launched_before_backward_returned"] >= 1, o             # overlap: buckets leave while the reverse pass is still running
    assert o["delta_equal"] and o["loss_a"] == o["loss_b"], o
    assert o["grad_max_abs_diff"] == 0.0, o                           # AVG over one rank: bit-identical gradients
    assert o["global_minmax_equals_plain_forward"], o                 # the 2-float MIN all-reduce of the optional global_minmax mode


def test_bench_self_launch_path_on_one_gpu():
    """the path the driver takes at N = 8 -- `python bench.py --gpus N` -> bench.self_launch ->
    `python -m torch.distributed.run` child -> RCCL process group -> bucketed gradient all-reduce -> rank 0's JSON line relayed by
    the parent -- run end to end on the one GPU of this box (PAIF_BENCH_FORCE_LAUNCH=1 takes the launcher path at N = 1, a 1-rank
    RCCL group runs the real collective).  Everything short of the 8-GPU node: exit status, exactly one JSON line, the number of ranks
    RCCL connected, the exposed all-reduce time of configs[4].  The parent of the launcher is a plain subprocess of a pytest process
    that has not initialised the GPU."""
    import torch

    if torch.cuda.is_initialized():
        msg = "the launcher is only started from a parent that has not touched the GPU"
        if os.environ.get("PAIF_REQUIRE_RCCL_TEST") == "1":
            pytest.fail(msg)
        pytest.skip(msg)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PAIF_BENCH_FORCE_LAUNCH="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "train", "--force-allreduce",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    o = json.loads(lines[0])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        open(os.path.join(out_dir, "bench_selflaunch.json"), "w").write(lines[0] + "\n")
    assert o["n_gpus"] == 1 and o["rccl_ranks_seen"] == 1, o
    assert o["steps"] == 1 and o["value"] > 0 and o["unit"] == "pairs/s" and o["scaling"] == "weak", o
    assert "configs[4]" in o["config"]["workload"], o["config"]
    assert o["allreduce_exposed_ms_per_step"] >= 0.0, o            # the bucketed all-reduce ran and its exposed tail was timed
    assert o["roofline"]["frac"] > 0 and o["roofline"]["launches"] > 0, o["roofline"]
