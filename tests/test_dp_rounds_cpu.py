"""CPU, world_size 2, gloo: the gradient exchange INSIDE backward (mmfusion.dp.BackwardExchange + the round flush of the
deferred wgrad queue, mmfusion.ops.set_wgrad_rounds).  A chain of six linear layers whose autograd
Functions queue their weight gradients exactly like the product Functions do; the weight-gradient GEMM itself is stood in
for by torch arithmetic (the product issues the grouped TN launch — test infrastructure here, as in test_dp_cpu.py).
Checked: rounds are delivered DURING backward from the second step on (the first has no history), a weight used twice is
held back for the final round, every element of the arena ends up as the mean over ranks — identical to the one-shot
all-reduce — for the f32 and the bf16 wire format."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, scenario="plain"):
    for p in (REPO, os.path.join(REPO, "simple-multimodal_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MMFUSION_CONFIG_MKDIRS"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmfusion import arena as arena_mod, dp, ops

    d, nlayer = 64, 6

    class FakeArena:                                     # what ops.queue_wgrad / dp need of a ParamArena, on the CPU
        def __init__(self):
            self.numel = nlayer * d * d + 128            # six (d, d) matrices + a tail of vector gradients
            self.grads = torch.zeros(self.numel)

        def take_first_touch(self, g):
            return False

        def w(self, i):
            return self.grads[i * d * d:(i + 1) * d * d].view(d, d)
    ar = FakeArena()
    arena_mod._ARENAS.add(ar)
    Ws = [torch.randn(d, d, generator=torch.Generator().manual_seed(10 + i)) * 0.2 for i in range(nlayer)]

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, i):
            ctx.i = i
            ctx.save_for_backward(x)
            return x @ Ws[i].t()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            ops.queue_wgrad(g, x, ar.w(ctx.i), None)     # dW = g^T x, deferred: exactly the product Functions' call
            return g @ Ws[ctx.i], None

    base_order = [0, 1, 2, 3, 4, 5, 2]                   # layer 2 is applied twice: two writers of one gradient region

    def order_of(step):
        """plain: the same chain every step.  diverge (round 4): step 3 — rank 1 alone drops layer 5 (as a branch
        switched off by modality dropout on one rank would); step 6 — both ranks use layer 5 a second time, at the head of the
        chain, unknown to the plan: its region is written again at the very end of backward, after its round went out."""
        if scenario == "diverge":
            if step == 3 and rank == 1:
                return [0, 1, 2, 3, 4, 2]
            if step == 6:
                return [5] + base_order                  # its backward comes LAST: layer 5's round went out long before
        return base_order

    order = base_order

    def fwd_bwd(x):
        nonlocal order
        h = x.clone().requires_grad_(True)
        y = h
        for i in order:
            y = torch.tanh(Lin.apply(y, i))
        y.pow(2).sum().backward()

    class Bx(dp.BackwardExchange):                       # the GEMM stand-in: dW (+)= dy^T x
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.log = []

        def _issue(self, problems):
            for (dy, x, wg, bg, _, overwrite) in problems:
                wg += dy.t() @ x

        def _on_round(self, problems, final):
            self.log.append((len(problems), final, in_backward[0]))
            super()._on_round(problems, final)

    in_backward = [False]
    results = {}
    for compress in (None, "bf16"):
        bx = Bx(ar, rounds=3, compress=compress).install()
        per_step = []
        for step in range(3 if scenario == "plain" else 8):
            order = order_of(step)
            x = torch.randn(8, d, generator=torch.Generator().manual_seed(1000 * step + rank))
            ar.grads.zero_()
            ar.grads[-128:] = float(rank + 1)            # something in the tail that only finish() can exchange
            bx.log.clear()
            in_backward[0] = True
            fwd_bwd(x)
            in_backward[0] = False
            bx.finish()
            got = ar.grads.clone()
            # this rank's local gradients from plain torch autograd on the same chain (the arena's early-round regions are
            # already being reduced in place by the time backward returns), then the one-shot exchange on them
            Wr = [w.clone().requires_grad_(True) for w in Ws]
            y = x
            for i in order:
                y = torch.tanh(y @ Wr[i].t())
            y.pow(2).sum().backward()
            local = torch.zeros(ar.numel)
            for i in range(nlayer):
                if Wr[i].grad is not None:               # (a layer this rank did not use leaves zeros)
                    local[i * d * d:(i + 1) * d * d] = Wr[i].grad.reshape(-1)
            local[-128:] = float(rank + 1)
            ref = type("A", (), {})()
            ref.grads = local
            dp.allreduce_grads(ref, compress=compress, bucket_bytes=1 << 20)
            per_step.append((list(bx.log), float((got - ref.grads).abs().max()), float(ref.grads.abs().max()),
                             float(got[-128:].mean()), bx.resends))
        bx.remove()
        results[str(compress)] = per_step
    q.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_inside_backward_equals_one_shot_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, results in res:
        for compress, per_step in results.items():
            for step, (log, err, scale, tail_mean, _resends) in enumerate(per_step):
                tol = 1e-5 if compress == "None" else 2 ** -7
                assert err <= tol * max(1.0, scale), f"rank {rank} {compress} step {step}: differs from the one-shot exchange by {err:.3e}"
                assert abs(tail_mean - 1.5) < 1e-2, "the tail (no wgrad) was not exchanged exactly once"
                assert log[-1][1] is True                               # the final round always comes from the end of backward
                if step == 0:
                    assert len(log) == 1 and log[0][0] == 7            # no history yet: everything in the final round
                else:
                    early = [e for e in log if not e[1]]
                    assert len(early) == 2 and all(e[2] for e in early), f"rounds were not delivered during backward: {log}"
                    # 7 problems, 3 rounds: cuts after 3 and 5 queued; the two writers of layer 2's gradient are held back
                    assert sum(e[0] for e in log) == 7 and log[-1][0] >= 2


def test_ranks_whose_graphs_differ_still_issue_the_same_collectives():
    """the in-backward exchange must not depend on every rank queueing the same weight-gradient problems in the same
    order.  Eight steps on two gloo ranks: in step 3 rank 1 alone drops a layer (fewer problems, fewer rounds), in step 6 both
    ranks use a layer twice that the agreed plan knows as a single-writer region (its round is on the wire when the second
    gradient arrives).  No step may hang, every step's arena must equal the one-shot mean, and step 6 must have exchanged the
    planned regions a second time."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "diverge")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, results in res:
        for compress, per_step in results.items():
            assert len(per_step) == 8
            for step, (log, err, scale, tail_mean, resends) in enumerate(per_step):
                tol = 1e-5 if compress == "None" else 2 ** -6
                assert err <= tol * max(1.0, scale), f"rank {rank} {compress} step {step}: differs from the one-shot exchange by {err:.3e}"
                assert abs(tail_mean - 1.5) < 1e-2
            assert per_step[5][4] == 0 and per_step[6][4] == 1, [s[4] for s in per_step]     # the late writer of step 6 was exchanged again
