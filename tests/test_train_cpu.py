"""CPU: host logic of the training-step harness (schedule, loss composition)."""
import torch

from mmfusion.train import fusion_loss, one_cycle, one_cycle_lr


def test_one_cycle_matches_torch():
    total, max_lr = 57, 3e-4
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=max_lr)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total, pct_start=0.1,
                                              anneal_strategy="cos")
    for step in range(total):
        assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(step, total, max_lr)) < 1e-12 + 1e-9 * max_lr, step
        opt.step()
        if step + 1 < total:
            sch.step()


def test_one_cycle_momentum_matches_torch_adam():
    """OneCycleLR's defaults (what the reference runs with, advanced_trainer.py:102-110) cycle Adam's beta1 between
    0.95 and 0.85 against the learning rate; ``one_cycle`` must reproduce both series."""
    total, max_lr = 40, 1e-3
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=max_lr)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total, pct_start=0.1, anneal_strategy="cos")
    seen = []
    for step in range(total):
        lr, b1 = one_cycle(step, total, max_lr)
        assert abs(opt.param_groups[0]["lr"] - lr) < 1e-12 + 1e-9 * max_lr, step
        assert abs(opt.param_groups[0]["betas"][0] - b1) < 1e-12, step
        assert opt.param_groups[0]["betas"][1] == 0.999
        seen.append(b1)
        opt.step()
        if step + 1 < total:
            sch.step()
    assert seen[0] == 0.95 and abs(min(seen) - 0.85) < 1e-12 and abs(seen[-1] - 0.95) < 1e-9


def test_fusion_loss_recipe():
    g = torch.Generator().manual_seed(0)
    logits, y = torch.randn(6, 7, generator=g), torch.randint(0, 7, (6,), generator=g)
    cl = {"a": torch.tensor(0.5), "b": torch.tensor(1.5)}
    want = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1) + 0.1 * 2.0
    assert torch.allclose(fusion_loss({"emotion_logits": logits, "contrastive_losses": cl}, y), want)
    assert torch.allclose(fusion_loss({"emotion_logits": logits, "contrastive_losses": {}}, y),
                          torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1))


def test_split_wgrad_by_offset_partitions_by_arena_range(monkeypatch):
    """Host logic of the overlapped data-parallel exchange (bench.py, N > 1): queued wgrad problems are cut into
    consecutive gradient-arena ranges of about equal GEMM work; problems writing one region stay in one part (the
    overwrite-then-accumulate order inside it is kept), the bounds are the first offsets of the parts."""
    from mmfusion import ops
    offs = {}

    def fake(q):
        return offs[id(q[2])]
    monkeypatch.setattr(ops, "wgrad_offset", fake)

    def prob(off, n_out, k_in, rows):
        dy, x, g = torch.empty(rows, n_out), torch.empty(rows, k_in), torch.empty(n_out, k_in)
        offs[id(g)] = off
        return (dy, x, g, None, None, True)
    # arena order: four equal matrices, one of them written twice (same region), then a small one
    a, b, b2, c, d, e = (prob(0, 64, 64, 100), prob(4096, 64, 64, 100), prob(4096, 64, 64, 100), prob(8192, 64, 64, 100),
                         prob(12288, 64, 64, 100), prob(16384, 8, 8, 100))
    pend = [e, c, b, a, d, b2]                              # queue order is backward order, not arena order
    parts, bounds = ops.split_wgrad_by_offset(pend, 2)
    assert len(parts) == 2 and bounds[0] == 0
    assert [offs[id(q[2])] for q in parts[0]] == sorted(offs[id(q[2])] for q in parts[0])
    got = [id(q) for p in parts for q in p]
    assert sorted(got) == sorted(id(q) for q in pend)       # a partition: nothing lost, nothing twice
    lo = {offs[id(q[2])] for q in parts[0]}
    hi = {offs[id(q[2])] for q in parts[1]}
    assert max(lo) < min(hi) == bounds[1]                   # consecutive ranges, bound = first offset of the second part
    assert (4096 in lo) != (4096 in hi)                     # both writers of the shared region on one side
    assert [id(q) for q in (parts[0] + parts[1]) if offs[id(q[2])] == 4096] == [id(b), id(b2)]   # queue order kept
    work = [sum(q[2].numel() * q[0].shape[0] for q in p) for p in parts]
    assert 0.3 < work[0] / sum(work) < 0.7
    # more parts than distinct regions: trailing parts are empty, bounds stay monotone
    parts3, bounds3 = ops.split_wgrad_by_offset([a], 3)
    assert [len(p) for p in parts3] == [1, 0, 0] and bounds3 == sorted(bounds3)


def test_wgrad_rounds_separate_writers_of_one_region():
    """A weight applied twice in one forward queues two wgrad problems for one gradient region; the grouped launch
    accumulates non-atomically, so they must land in different launch rounds, in queue order."""
    from mmfusion import ops
    g1, g2, b1 = torch.empty(8, 4), torch.empty(8, 4), torch.empty(8)
    dy, x = torch.empty(5, 8), torch.empty(5, 4)
    a = (dy, x, g1, b1, None, True)          # first touch of g1 (overwrite)
    b = (dy, x, g2, None, None, True)
    c = (dy, x, g1, b1, None, False)         # second use of the same Linear
    d = (dy, x, g1[2:6], None, None, False)  # a row block of the same region (overlap, different pointer)
    e = (dy, x, torch.empty(3, 3), b1, None, False)   # other weight, same bias vector
    rounds = ops._wgrad_rounds([a, b, c, d, e])
    assert [len(r) for r in rounds] == [2, 1, 2]          # d and e touch disjoint memory: one round
    assert rounds[0] == [a, b] and rounds[1] == [c] and rounds[2] == [d, e]
