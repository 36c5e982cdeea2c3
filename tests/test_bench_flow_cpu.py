"""CPU: control flow of bench.py for N > 1.

Both round-1 hangs (gpurun_out/b17.log, b48_gloo2_train.log) had one cause: after the timed region rank 0 alone ran a
per-kernel profiling pass of the training step, which contained the gradient all-reduce, while the other ranks had
moved on to the final barrier.  bench.py now makes that impossible by construction: every rank runs
``timed_region`` (collectives included) and then ``leave_group`` (barrier + destroy_process_group); rank 0's
profiling / CPU baseline / printing come after that, where mmfusion.dp's exchanges see a world of one and return.
Here: two gloo ranks on the CPU run exactly that sequence with a stand-in step."""
import os
import subprocess
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (REPO, os.path.join(REPO, "simple-multimodal_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import bench
    from mmfusion import dp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grads = torch.full((1000,), float(rank + 1))
    calls = []

    def step():                                   # stand-in for one bench step: compute + the gradient exchange
        g = grads.clone()
        dp.allreduce_flat(g, average=True, bucket_bytes=1024)
        calls.append(float(g[0]))
    elapsed = bench.timed_region(step, steps=3, warmup=1, world=world, dist=dist)
    bench.leave_group(world, dist)
    still = dist.is_initialized()
    post = None
    if rank == 0:                                 # the rank-0-only tail: a step INCLUDING the exchange must not wait for rank 1
        g = grads.clone()
        dp.allreduce_flat(g, average=True)
        fake = SimpleNamespace(grads=grads.clone())
        dp.allreduce_grads(fake, compress="bf16")
        h = dp.allreduce_grads_range_async(fake, 0, 500, compress=None)
        h.finish()
        post = (float(g[0]), float(fake.grads[0]))
    q.put((rank, elapsed, calls, still, post))


def test_two_gloo_ranks_leave_the_group_before_rank0_only_work():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 0                      # max-over-ranks time: the same number on both ranks
    for rank, elapsed, calls, still, post in res:
        assert calls == [1.5] * 4                          # 1 warm-up + 3 timed steps, each a mean of 1 and 2
        assert still is False                              # the group is gone before any rank-0-only code
    assert res[0][4] == (1.0, 1.0) and res[1][4] is None   # rank 0's tail ran alone: exchanges were no-ops, no hang


def test_direct_multi_gpu_invocation_exits_cleanly_without_gpus():
    """`python bench.py --gpus 2` is no longer a bare SystemExit asking for torchrun: it launches the ranks itself, or —
    with fewer GPUs than ranks, as in this container — exits non-zero with a message, without hanging."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_self_launch_starts_the_ranks_as_children_and_relays_the_exit_code():
    """With the gloo backend the device-count gate does not apply, so the self-launcher really starts two ranks
    (torch.distributed.run as a child process); on a box without a GPU each rank exits with the bench's own
    "needs an MI355X" message and the launcher's non-zero code comes back."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        return                                             # on a GPU box this would run the real rehearsal: not a CPU test
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr
