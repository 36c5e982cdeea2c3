"""CPU: host logic of the training-step harness (schedule, loss composition)."""
import torch

from mmfusion.train import fusion_loss, one_cycle_lr


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


def test_fusion_loss_recipe():
    g = torch.Generator().manual_seed(0)
    logits, y = torch.randn(6, 7, generator=g), torch.randint(0, 7, (6,), generator=g)
    cl = {"a": torch.tensor(0.5), "b": torch.tensor(1.5)}
    want = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1) + 0.1 * 2.0
    assert torch.allclose(fusion_loss({"emotion_logits": logits, "contrastive_losses": cl}, y), want)
    assert torch.allclose(fusion_loss({"emotion_logits": logits, "contrastive_losses": {}}, y),
                          torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1))
