"""Training-step harness for the fusion path.

Reproduces the recipe of the reference's ``AdvancedTrainer`` for the part of the model this build owns
(reference ``training/advanced_trainer.py``):

  * loss (:139-166)      CrossEntropy(label_smoothing=0.1) on ``emotion_logits`` + 0.1 * sum of the
                         contrastive losses (+ 0.1 * aux, + 0.5 * distillation when present);
  * optimiser (:85-94)   AdamW, weight_decay 1e-5; the reference's 0.1x LR group only holds HF-backbone
                         parameters, which are outside this build, so all fusion parameters use ``lr``;
  * schedule (:102-110)  OneCycleLR(max_lr, pct_start=0.1, cos);
  * clipping (:174,179)  ``clip_grad_norm_(1.0)``;
  * no per-step ``.item()`` syncs (the reference does four, :185-188).

MI355X-native part: ``FusedAdamW`` runs two HIP kernels over the flat arenas — a sum-of-squares
reduction and one fused clip + AdamW pass that updates the fp32 masters *and* the bf16 shadow the MFMA
GEMMs read, so a training step has no separate weight-cast kernel.  Hyper-parameters sit in a device
array, so the whole step (forward, backward, optimiser) can be replayed from one hipGraph.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dp, lib
from .arena import ParamArena


def one_cycle_lr(step: int, total_steps: int, max_lr: float, pct_start: float = 0.1,
                 div_factor: float = 25.0, final_div_factor: float = 1e4) -> float:
    """torch.optim.lr_scheduler.OneCycleLR (anneal_strategy='cos', three_phase=False): LR to use for
    optimiser step number ``step`` (0-based: step 0 uses initial_lr = max_lr / div_factor)."""
    initial, minimum = max_lr / div_factor, max_lr / div_factor / final_div_factor
    up_end = float(pct_start * total_steps) - 1.0
    down_end = float(total_steps) - 1.0
    cos = lambda a, b, pct: b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    if step <= up_end or up_end >= down_end:
        return cos(initial, max_lr, step / up_end if up_end > 0 else 1.0)
    return cos(max_lr, minimum, min(1.0, (step - up_end) / (down_end - up_end)))


class FusedAdamW:
    """AdamW + global-norm clipping over a ``ParamArena`` (all parameters in one launch)."""

    def __init__(self, arena: ParamArena, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-5, max_grad_norm: Optional[float] = 1.0):
        self.arena, self.lr, self.betas, self.eps = arena, lr, betas, eps
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        dev = arena.master.device
        self.exp_avg = torch.zeros_like(arena.master)
        self.exp_avg_sq = torch.zeros_like(arena.master)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hparams = torch.zeros(9, dtype=torch.float32, device=dev)
        self._hp_host = torch.zeros(9, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(9)
        self.t = 0

    def set_hparams(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        """Advance the step counter and upload this step's hyper-parameters (call OUTSIDE a captured
        graph, before replaying it)."""
        self.t += 1
        b1, b2 = self.betas
        h = self._hp_host
        h[0], h[1], h[2], h[3], h[4] = (self.lr if lr is None else lr), b1, b2, self.eps, self.weight_decay
        h[5], h[6] = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        h[7] = self.max_grad_norm if self.max_grad_norm else 0.0
        h[8] = grad_scale
        self.hparams.copy_(h, non_blocking=True)

    def launch(self) -> None:
        """Enqueue norm + update (graph-capturable; uses the hyper-parameters currently on the device)."""
        a, L, st = self.arena, lib.load(), lib.stream_ptr()
        gn = None
        if self.max_grad_norm:
            self.gnorm_sq.zero_()
            lib.check(L.mmf_sqnorm_f32(a.grads.data_ptr(), a.numel, self.gnorm_sq.data_ptr(), st))
            gn = self.gnorm_sq.data_ptr()
        lib.check(L.mmf_adamw_step(a.master.data_ptr(), a.grads.data_ptr(), self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), a.shadow.data_ptr(), a.numel,
                                   self.hparams.data_ptr(), gn, st))
        a.mark_shadow_fresh()

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        self.set_hparams(lr, grad_scale)
        self.launch()


def fusion_loss(outputs: Dict, targets: torch.Tensor, label_smoothing: float = 0.1) -> torch.Tensor:
    """reference advanced_trainer.py:139-166 (aux terms only when the batch carries them — the
    reference's ``hasattr(batch, 'valence')`` on a dict is always False)."""
    loss = F.cross_entropy(outputs["emotion_logits"], targets, label_smoothing=label_smoothing)
    cl = outputs.get("contrastive_losses") or {}
    if cl:
        loss = loss + 0.1 * sum(cl.values())
    if "distillation_loss" in outputs:
        loss = loss + 0.5 * outputs["distillation_loss"]
    return loss


class FusionTrainStep:
    """fusion module + classifier head + loss + (RCCL all-reduce) + fused AdamW, as one callable.

    ``model(text, audio, video, compute_contrastive_loss=...)`` must return a dict with
    ``fused_features``; ``head`` maps them to emotion logits (reference ``EmotionClassifier``)."""

    def __init__(self, model: torch.nn.Module, head: torch.nn.Module, arena: ParamArena, *, lr: float = 1e-4,
                 weight_decay: float = 1e-5, max_grad_norm: float = 1.0, total_steps: int = 1000,
                 contrastive: bool = True, allreduce: Optional[str] = "bf16"):
        self.model, self.head, self.arena = model, head, arena
        self.opt = FusedAdamW(arena, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.max_lr, self.total_steps, self.contrastive, self.allreduce = lr, total_steps, contrastive, allreduce
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    def fwd_bwd(self, text, audio, video, targets) -> torch.Tensor:
        self.arena.zero_grad(overlap=True, lazy=True)
        kw = {"compute_contrastive_loss": True} if self.contrastive else {}
        out = self.model(text, audio, video, **kw)
        fused = out["fused_features"] if isinstance(out, dict) else out
        outputs = dict(out) if isinstance(out, dict) else {}
        outputs["emotion_logits"] = self.head(fused)
        loss = fusion_loss(outputs, targets)
        loss.backward()
        self.arena.finalize_grads()
        return loss

    def __call__(self, text, audio, video, targets) -> torch.Tensor:
        loss = self.fwd_bwd(text, audio, video, targets)
        if self.world > 1:
            dp.allreduce_grads(self.arena, compress=None if self.allreduce == "fp32" else "bf16")
        self.opt.step(lr=one_cycle_lr(self.opt.t, self.total_steps, self.max_lr))
        return loss
