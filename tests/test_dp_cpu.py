"""CPU, world_size 2, gloo: the data-parallel exchange (mmfusion/dp.py).  The compute stand-in is the
oracle (test infrastructure); what is under test is the sharding + bucketed mean all-reduce logic:
two ranks on half batches must end up with exactly the full-batch gradient on both ranks."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmfusion import dp, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat_grads(P, xs, H):
    from oracle import ref_cpu
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = ref_cpu.multimodal_transformer(Pg, "", *xs, H)
    out["fused_features"].sum().backward()
    return torch.cat([Pg[k].grad.reshape(-1) for k in sorted(Pg)])


def _worker(rank, world, port, q):
    for p in (REPO, os.path.join(REPO, "simple-multimodal_amd"), os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from helpers import load_fixture, fixture_params
    meta = load_fixture("mult_seq_dh64").meta
    P = fixture_params(meta)
    B = 4
    xs = synth.make_features(B, meta["Ts"], meta["d"], seed=77)
    full = _flat_grads(P, xs, 2)                                   # sum-loss over the global batch
    mine = _flat_grads(P, [dp.shard_batch(x, rank, world) for x in xs], 2)
    # sum-loss: the global gradient is the SUM of the shard gradients -> average=False;
    # tiny buckets force several collectives and a ragged last bucket
    dp.allreduce_flat(mine, average=False, bucket_bytes=100_000)
    err = float((mine - full).abs().max() / full.abs().max())
    avg = torch.full((1000,), float(rank + 1))
    dp.allreduce_flat(avg, average=True, bucket_bytes=1024)        # mean of 1 and 2
    # bf16-compressed path on an arena-shaped object: mean of the two shard gradients, 2^-8 accurate
    from types import SimpleNamespace
    shard = _flat_grads(P, [dp.shard_batch(x, rank, world) for x in xs], 2)
    fake = SimpleNamespace(grads=shard.clone())
    dp.allreduce_grads(fake, compress="bf16", bucket_bytes=100_000)
    cerr = float((fake.grads - full / world).abs().max() / (full / world).abs().max())
    # range-wise asynchronous form (what bench.py overlaps with the second half of the wgrad flush): two ranges
    # started one after the other, finished in order, must give the same result as the one-shot call, both formats
    n, cut = shard.numel(), (shard.numel() // 3) // 64 * 64
    rerr = []
    for compress in (None, "bf16"):
        one = SimpleNamespace(grads=shard.clone())
        dp.allreduce_grads(one, compress=compress, bucket_bytes=100_000)
        two = SimpleNamespace(grads=shard.clone())
        hA = dp.allreduce_grads_range_async(two, 0, cut, compress=compress, bucket_bytes=100_000)
        hB = dp.allreduce_grads_range_async(two, cut, n, compress=compress, bucket_bytes=100_000)
        hA.finish()
        hB.finish()
        rerr.append(float((two.grads - one.grads).abs().max()))
    q.put((rank, err, float(avg.min()), float(avg.max()), cerr, max(rerr)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, lo, hi, cerr, rerr in res:
        assert err < 1e-5, f"rank {rank}: all-reduced gradient deviates by {err}"
        assert lo == hi == 1.5
        assert cerr < 2 ** -6, f"rank {rank}: bf16-compressed all-reduce deviates by {cerr}"
        assert rerr == 0.0, f"rank {rank}: range-wise all-reduce differs from the one-shot call by {rerr}"


def test_bucket_bounds_cover_exactly():
    for n, es, bb in [(1, 4, 64), (1000, 4, 1024), (51_380_000, 4, 64 << 20), (130, 2, 256)]:
        b = dp.bucket_bounds(n, es, bb)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
        assert all((e - s) * es <= max(bb, 64 * es) for s, e in b)


def test_shard_batch_rejects_ragged():
    with pytest.raises(ValueError):
        dp.shard_batch(torch.zeros(5, 3), 0, 2)
