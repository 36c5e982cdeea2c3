"""Training-step harness for the fusion path.

Reproduces the recipe of the reference's ``AdvancedTrainer`` for the part of the model this build owns
(reference ``training/advanced_trainer.py``):

  * loss (:139-166)      CrossEntropy(label_smoothing=0.1) on ``emotion_logits`` + 0.1 * sum of the
                         contrastive losses (+ 0.1 * aux, + 0.5 * distillation when present);
  * optimiser (:85-94)   AdamW, weight_decay 1e-5; the reference's 0.1x LR group only holds HF-backbone
                         parameters, which are outside this build, so all fusion parameters use ``lr``;
  * schedule (:102-110)  OneCycleLR(max_lr, pct_start=0.1, cos) with torch's defaults, i.e. INCLUDING
                         ``cycle_momentum``: Adam's beta1 runs 0.95 -> 0.85 -> 0.95 against the learning rate;
  * clipping (:174,179)  ``clip_grad_norm_(1.0)``;
  * no per-step ``.item()`` syncs (the reference does four, :185-188).

MI355X-native part: ``FusedAdamW`` runs two HIP kernels over the flat arenas — a sum-of-squares
reduction and one fused clip + AdamW pass that updates the fp32 masters *and* the bf16 shadow the MFMA
GEMMs read, so a training step has no separate weight-cast kernel.  Hyper-parameters sit in a device
array, so the whole step (forward, backward, optimiser) can be replayed from one hipGraph.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional

import torch
import torch.nn.functional as F

from . import dp, lib
from .arena import ParamArena


def one_cycle(step: int, total_steps: int, max_lr: float, pct_start: float = 0.1, div_factor: float = 25.0,
              final_div_factor: float = 1e4, base_momentum: float = 0.85, max_momentum: float = 0.95):
    """torch.optim.lr_scheduler.OneCycleLR (anneal_strategy='cos', three_phase=False, cycle_momentum=True — the
    defaults the reference runs with, advanced_trainer.py:102-110): (learning rate, Adam beta1) to use for optimiser
    step number ``step`` (0-based: step 0 uses initial_lr = max_lr / div_factor and beta1 = max_momentum)."""
    initial, minimum = max_lr / div_factor, max_lr / div_factor / final_div_factor
    up_end = float(pct_start * total_steps) - 1.0
    down_end = float(total_steps) - 1.0
    anneal = lambda a, b, pct: b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    if step <= up_end or up_end >= down_end:
        pct = step / up_end if up_end > 0 else 1.0
        return anneal(initial, max_lr, pct), anneal(max_momentum, base_momentum, pct)
    pct = min(1.0, (step - up_end) / (down_end - up_end))
    return anneal(max_lr, minimum, pct), anneal(base_momentum, max_momentum, pct)


def one_cycle_lr(step: int, total_steps: int, max_lr: float, pct_start: float = 0.1,
                 div_factor: float = 25.0, final_div_factor: float = 1e4) -> float:
    """The learning-rate half of ``one_cycle``."""
    return one_cycle(step, total_steps, max_lr, pct_start, div_factor, final_div_factor)[0]


class FusedAdamW:
    """AdamW + global-norm clipping over a ``ParamArena`` (all parameters in one launch).

    Hyper-parameters live in a device array.  Two ways to advance a step:

      * ``advance()`` — device-side (graph-capturable): one thread of ``mmf_adamw_advance`` increments the
        device-resident step counter and derives the bias corrections and, with ``set_schedule``, the OneCycle
        learning rate.  Nothing crosses the PCIe bus per step, so a captured step replays correctly however far
        the host runs ahead.  The DEVICE counter ``step_dev`` is the source of truth: ``self.t`` is only the host's
        running guess (a captured ``advance()`` replayed K times moves ``step_dev`` by K and ``self.t`` by one), so
        every consumer of the step count — ``state_dict``, ``set_hparams`` — first reads it back (``sync_step``).
      * ``set_hparams(lr)`` — host-side, for a learning rate the host computes: the values go through a RING of
        pinned buffers, each guarded by an event recorded behind its copy, so a buffer is never rewritten while
        an asynchronous copy may still read it (the advisor's round-1 finding: one pinned buffer rewritten every
        step is read when the GPU executes the copy, not when the host enqueues it)."""

    _RING = 32

    def __init__(self, arena: ParamArena, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-5, max_grad_norm: Optional[float] = 1.0, shard: bool = False, group=None):
        """``shard=True`` under an initialised process group of N > 1 ranks: the ZeRO-1 form of the step (``launch`` then
        runs ``launch_sharded``): reduce-scatter of the gradient arena instead of an all-reduce, clip + AdamW on this
        rank's 1/N of the arena only (the moments exist for that shard only), all-gather of the bf16 shadow the forward
        reads, plus the fp32 masters of the small parameters (the arena's tail).  The fp32 masters of the big matrices of the
        OTHER shards go stale: ``gather_masters()`` refreshes them (checkpoints do).  Not for the fp32 parity mode."""
        self.arena, self.lr, self.betas, self.eps = arena, lr, betas, eps
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        dev = arena.master.device
        self.group = group
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank(group) if self.world > 1 else 0
        self.sharded = bool(shard) and self.world > 1
        if self.sharded:
            from . import ops as _ops
            if _ops.fp32_mode():
                raise RuntimeError("FusedAdamW(shard=True): the fp32 parity mode reads the fp32 MASTERS as weights, and the sharded "
                                   "step keeps only this rank's shard of them up to date — use the replicated optimiser there")
            if arena.capacity % (64 * self.world):
                raise ValueError(f"arena capacity {arena.capacity} does not divide into {self.world} 64-aligned shards")
            self.shard_len = arena.capacity // self.world
            self.shard_start = self.rank * self.shard_len
        else:
            self.shard_len, self.shard_start = arena.numel, 0
        self._tail = (torch.zeros(max(0, arena.numel - arena.small_start), dtype=torch.float32, device=dev)
                      if self.sharded else None)         # the small-parameter exchange buffer of launch_sharded, allocated once
        self._grad_scale_base = 1.0                      # the caller's grad_scale (set_hparams): the sharded step multiplies 1/world in
        self.exp_avg = torch.zeros(self.shard_len, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.shard_len, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hparams = torch.zeros(9, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.sched = torch.zeros(9, dtype=torch.float64, device=dev)
        self._cuda = dev.type == "cuda"
        self._ring = [torch.zeros(9, dtype=torch.float32).pin_memory() if self._cuda else torch.zeros(9)
                      for _ in range(self._RING)]
        self._ring_events = [None] * self._RING
        self._ring_pos = 0
        self.t = 0
        self._dev_advanced = False           # an advance() was enqueued since the last read-back
        self._captured = False               # an advance() sits in a captured graph: step_dev may move at any replay
        self._upload(self._static_hparams(lr, 1.0))

    def _static_hparams(self, lr: float, grad_scale: float, beta1: Optional[float] = None):
        b1, b2 = self.betas
        b1 = b1 if beta1 is None else beta1
        return [lr, b1, b2, self.eps, self.weight_decay, 1.0 - b1 ** max(self.t, 1), 1.0 - b2 ** max(self.t, 1),
                self.max_grad_norm if self.max_grad_norm else 0.0, grad_scale]

    def _upload(self, values) -> None:
        i = self._ring_pos
        self._ring_pos = (i + 1) % self._RING
        ev = self._ring_events[i]
        if ev is not None:
            ev.synchronize()                 # the copy that last read this buffer has executed (normally long ago)
        h = self._ring[i]
        for k, v in enumerate(values):
            h[k] = v
        self.hparams.copy_(h, non_blocking=True)
        if self._cuda:
            ev = ev or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._ring_events[i] = ev

    def set_schedule(self, max_lr: float, total_steps: int, pct_start: float = 0.1, div_factor: float = 25.0,
                     final_div_factor: float = 1e4, cycle_momentum: bool = True, base_momentum: float = 0.85,
                     max_momentum: float = 0.95) -> None:
        """OneCycleLR(cos) evaluated on the device by ``advance()``.  Defaults are torch's, which the reference
        runs with (advanced_trainer.py:102-110): that includes ``cycle_momentum=True`` — Adam's beta1 is cycled
        0.95 -> 0.85 -> 0.95 against the learning rate."""
        self.sched.copy_(torch.tensor([1.0, max_lr, float(total_steps), pct_start, div_factor, final_div_factor,
                                       1.0 if cycle_momentum else 0.0, base_momentum, max_momentum], dtype=torch.float64))

    def sync_step(self) -> int:
        """Read the device-resident step counter back into ``self.t`` (one host sync: call at checkpoint time or
        before a host-side ``set_hparams``, never per step).  Not callable while a stream is capturing."""
        if self._cuda and (self._dev_advanced or self._captured) and not torch.cuda.is_current_stream_capturing():
            self.t = int(self.step_dev.item())
            self._dev_advanced = False
        return self.t

    def advance(self) -> None:
        """Device-side step advance (graph-capturable): counter += 1, bias corrections, scheduled LR."""
        self.t += 1
        self._dev_advanced = True
        if self._cuda and torch.cuda.is_current_stream_capturing():
            self._captured = True            # replays of the captured launch move step_dev without this code running
        lib.check(lib.load().mmf_adamw_advance(self.step_dev.data_ptr(), self.hparams.data_ptr(),
                                               self.sched.data_ptr(), lib.stream_ptr()))

    def set_hparams(self, lr: Optional[float] = None, grad_scale: float = 1.0, beta1: Optional[float] = None) -> None:
        """Host-side step advance: upload this step's hyper-parameters (call OUTSIDE a captured graph, before
        replaying it).  Safe against host run-ahead (ring of event-guarded pinned buffers).  ``beta1``: this step's
        Adam beta1 when the schedule cycles it (``one_cycle``); the bias correction uses it, as torch's Adam does."""
        self.sync_step()                     # graph replays of advance() moved the device counter, not self.t
        self.t += 1
        self._grad_scale_base = float(grad_scale)
        self._upload(self._static_hparams(self.lr if lr is None else lr, grad_scale, beta1))
        if self._cuda:
            self.step_dev.fill_(self.t)      # keep the device counter in step for a later advance()

    # -- the two kernels, on elements [s, s + n) of the arena (moments indexed from the shard's start) ------------------
    def _sqnorm_range(self, s: int, n: int) -> None:
        a = self.arena
        lib.check(lib.load().mmf_sqnorm_f32(a.grads_full.data_ptr() + 4 * s, n, self.gnorm_sq.data_ptr(), lib.stream_ptr()))

    def _adamw_range(self, s: int, n: int, use_norm: bool) -> None:
        a = self.arena
        lib.check(lib.load().mmf_adamw_step(a.master_full.data_ptr() + 4 * s, a.grads_full.data_ptr() + 4 * s,
                                            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                            a.shadow_full.data_ptr() + 2 * s, n, self.hparams.data_ptr(),
                                            self.gnorm_sq.data_ptr() if use_norm else None, lib.stream_ptr()))

    def launch(self) -> None:
        """Enqueue norm + update (graph-capturable when not sharded; uses the hyper-parameters currently on the device)."""
        if self.sharded:
            self.launch_sharded()
            return
        a = self.arena
        if self.max_grad_norm:
            self.gnorm_sq.zero_()
            self._sqnorm_range(0, a.numel)
        self._adamw_range(0, a.numel, bool(self.max_grad_norm))
        a.mark_shadow_fresh()

    def launch_sharded(self, compress: Optional[str] = None) -> None:
        """ZeRO-1 step (SURVEY 8e / VERDICT r2 item 5b).  In: every rank's LOCAL gradient sums in ``arena.grads`` (no
        all-reduce has run).  (1) in-place reduce-scatter (SUM) of the padded gradient arena: this rank's shard now holds
        the sum over ranks — N-1/N of the arena's bytes leave each GPU once, against twice for an all-reduce; the mean's 1/N
        rides in the kernel's gradient scale; (2) squared norm of the shard, summed over ranks (one scalar all-reduce): the
        global norm the clip needs; (3) fused clip + AdamW on the shard: masters, moments, bf16 shadow; (4) in-place
        all-gather of the bf16 shadow (2 B/param); (5) the fp32 masters of the small-parameter tail, ~1 MB.  Unmeasured on
        RCCL hardware."""
        import torch.distributed as dist
        a, n, s = self.arena, self.shard_len, self.shard_start
        mine = a.grads_full[s:s + n]
        dist.reduce_scatter_tensor(mine, a.grads_full, op=dist.ReduceOp.SUM, group=self.group)
        # the kernels scale every gradient by hparams[8] (grad_scale): the caller's scale (loss scaling, gradient accumulation;
        # set_hparams) times the mean over ranks — multiplied in from the saved base, not overwritten
        self.hparams[8:9].fill_(self._grad_scale_base / self.world)
        if self.max_grad_norm:
            self.gnorm_sq.zero_()
            self._sqnorm_range(s, n)                         # of the SUMMED shard: the kernel applies the 1/N itself
            dist.all_reduce(self.gnorm_sq, op=dist.ReduceOp.SUM, group=self.group)
        self._adamw_range(s, n, bool(self.max_grad_norm))
        dist.all_gather_into_tensor(a.shadow_full, a.shadow_full[s:s + n], group=self.group)
        # The small parameters (biases, LayerNorm vectors, attention vectors, the narrow heads: the arena's tail from
        # `small_start`) are read by the kernels as fp32 MASTERS, not through the shadow: bring them up to date on every
        # rank — each rank contributes the part of the tail it owns, zeros elsewhere, one sum all-reduce of ~1 MB.
        t0, t1 = a.small_start, a.numel
        if t1 > t0:
            tmp = self._tail
            tmp.zero_()
            lo, hi = max(t0, s), min(t1, s + n)
            if hi > lo:
                tmp[lo - t0:hi - t0].copy_(a.master_full[lo:hi])
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            a.master_full[t0:t1].copy_(tmp)
        # restore the caller's scale on the device (a later replicated launch, or a host that reads hparams back, sees it)
        self.hparams[8:9].fill_(self._grad_scale_base)
        a.mark_shadow_fresh()
        # from here on this rank's fp32 masters OUTSIDE its shard (and outside the small tail) are one step behind: every reader
        # of the masters — arena.refresh() when it would really cast, dp.broadcast_params, module.state_dict() — gathers first
        # (or raises where a collective cannot be assumed); gather_masters() clears the flag
        a.masters_stale = self

    def gather_masters(self) -> None:
        """Sharded mode: bring every rank's fp32 masters up to date (all-gather, 4 B/param) — before a checkpoint, an
        evaluation in fp32 mode, or leaving sharded mode.  No-op otherwise."""
        if self.sharded:
            import torch.distributed as dist
            a, n, s = self.arena, self.shard_len, self.shard_start
            dist.all_gather_into_tensor(a.master_full, a.master_full[s:s + n].clone(), group=self.group)
            a.masters_stale = None

    def _full_moments(self):
        """(exp_avg, exp_avg_sq) over the whole arena: the shard's in replicated mode, gathered in sharded mode."""
        if not self.sharded:
            return self.exp_avg, self.exp_avg_sq
        import torch.distributed as dist
        out = []
        for t in (self.exp_avg, self.exp_avg_sq):
            full = torch.empty(self.arena.capacity, dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t, group=self.group)
            out.append(full)
        return out[0], out[1]

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0, beta1: Optional[float] = None) -> None:
        self.set_hparams(lr, grad_scale, beta1)
        self.launch()

    # -- checkpoint interchange ------------------------------------------------------------------
    def state_dict(self, params: Iterable[torch.nn.Parameter]) -> Dict:
        """The moments in ``torch.optim.AdamW.state_dict()`` layout (what the reference's checkpoints hold,
        advanced_trainer.py:396-404), one parameter group, parameters numbered in the order of ``params``
        (pass ``module.parameters()``: the order ``AdamW(module.parameters())`` numbers them in).  Tensors are
        copies on the CPU."""
        a = self.arena
        self.sync_step()                     # the device counter is the truth (captured advance() replays)
        where = {id(p): i for i, p in enumerate(a.params)}
        state, ids = {}, []
        m1, m2 = self._full_moments()        # (sharded mode: a collective — every rank must call state_dict)
        for j, p in enumerate(params):
            i = where[id(p)]
            o, n = a.offsets[i], p.numel()
            ids.append(j)
            state[j] = {"step": torch.tensor(float(self.t)),
                        "exp_avg": m1[o:o + n].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": m2[o:o + n].view(p.shape).detach().cpu().clone()}
        if len(ids) != len(a.params):
            raise ValueError("FusedAdamW.state_dict: `params` must enumerate exactly the arena's parameters")
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": True, "params": ids}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict, params: Iterable[torch.nn.Parameter]) -> None:
        """Inverse of ``state_dict``; also accepts a ``torch.optim.AdamW`` state over the same parameter order
        with several parameter groups (the reference's two learning-rate groups, advanced_trainer.py:85-94):
        only the moments and the step count are taken, the hyper-parameters stay this object's."""
        a = self.arena
        where = {id(p): i for i, p in enumerate(a.params)}
        order = [q for g in sd["param_groups"] for q in g["params"]]
        plist = list(params)
        if len(order) != len(plist) or len(plist) != len(a.params):
            raise ValueError(f"FusedAdamW.load_state_dict: checkpoint has {len(order)} parameters, the arena {len(a.params)}")
        steps = set()
        with torch.no_grad():
            dev = self.exp_avg.device
            m1 = torch.zeros(a.capacity, dtype=torch.float32, device=dev) if self.sharded else self.exp_avg.zero_()
            m2 = torch.zeros(a.capacity, dtype=torch.float32, device=dev) if self.sharded else self.exp_avg_sq.zero_()
            for key, p in zip(order, plist):
                st = sd["state"].get(key)
                if st is None:                        # a parameter that never received a gradient
                    continue
                i = where[id(p)]
                o, n = a.offsets[i], p.numel()
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdamW.load_state_dict: parameter {key}: moment shape "
                                     f"{tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
                m1[o:o + n].copy_(st["exp_avg"].reshape(-1))
                m2[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if self.sharded:                          # keep this rank's shard of the full moments
                s0 = self.shard_start
                self.exp_avg.copy_(m1[s0:s0 + self.shard_len])
                self.exp_avg_sq.copy_(m2[s0:s0 + self.shard_len])
        if len(steps) > 1:
            raise ValueError(f"FusedAdamW.load_state_dict: parameters at different step counts {sorted(steps)}: "
                             "the fused kernel applies one bias correction to the whole arena")
        self.t = steps.pop() if steps else 0
        self.step_dev.fill_(self.t)
        self._dev_advanced = False


def save_checkpoint(path: str, module: torch.nn.Module, optimizer: Optional["FusedAdamW"] = None, *, epoch: int = 0,
                    metrics: Optional[Dict] = None, config=None, scheduler_state: Optional[Dict] = None) -> None:
    """Write the reference's checkpoint layout (advanced_trainer.py:396-411): epoch, model_state_dict,
    optimizer_state_dict, scheduler_state_dict, metrics, config.  ``module.state_dict()`` has the reference's keys
    and shapes, so the reference's ``load_pretrained_model`` / ``load_state_dict`` read the file as their own."""
    if optimizer is not None and optimizer.sharded:
        # Sharded optimiser: gathering the other ranks' shards of the masters and moments is a COLLECTIVE — every rank of the group
        # must call save_checkpoint (and only rank 0 needs to pass a path it really writes to).  The usual rank-0-only save of
        # the reference trainer would wait for ranks that never arrive; the barrier below turns that into an error after
        # MMF_CKPT_BARRIER_S seconds (default 120) instead of a silent hang.
        import datetime
        import os as _os
        import torch.distributed as dist
        work = dist.barrier(group=optimizer.group, async_op=True)
        if not work.wait(timeout=datetime.timedelta(seconds=float(_os.environ.get("MMF_CKPT_BARRIER_S", "120")))):
            raise RuntimeError("save_checkpoint with a sharded optimiser is a collective: call it on EVERY rank of the group")
        optimizer.gather_masters()
    ckpt = {"epoch": int(epoch),
            "model_state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()},
            "optimizer_state_dict": optimizer.state_dict(module.parameters()) if optimizer is not None else {},
            "scheduler_state_dict": dict(scheduler_state or {}),
            "metrics": dict(metrics or {}),
            "config": config}
    torch.save(ckpt, path)


def load_checkpoint(path: str, module: torch.nn.Module, optimizer: Optional["FusedAdamW"] = None,
                    arena: Optional[ParamArena] = None) -> Dict:
    """Restore ``module`` (and the fused optimiser's moments / step count) from a checkpoint in the reference's
    layout, read with the safe loader (``models.multimodal_model.load_checkpoint_file``).  The parameters live in
    the arena's fp32 master, which ``load_state_dict`` writes in place; the bf16 shadow is re-cast.  Returns the
    checkpoint dict (epoch, metrics, scheduler_state_dict, config)."""
    from models.multimodal_model import load_checkpoint_file
    ckpt = load_checkpoint_file(path)
    module.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer_state_dict"):
        optimizer.load_state_dict(ckpt["optimizer_state_dict"], module.parameters())
    ar = arena if arena is not None else (optimizer.arena if optimizer is not None else None)
    if ar is not None:
        ar.refresh(force=True)
    return ckpt


def fusion_loss(outputs: Dict, targets: torch.Tensor, label_smoothing: float = 0.1) -> torch.Tensor:
    """reference advanced_trainer.py:139-166 (aux terms only when the batch carries them — the
    reference's ``hasattr(batch, 'valence')`` on a dict is always False)."""
    logits = outputs["emotion_logits"]
    cl = outputs.get("contrastive_losses") or {}
    if logits.is_cuda and logits.dim() == 2 and logits.shape[1] <= 64 and targets.dim() == 1:
        # one launch for the value and the gradient with respect to the logits (csrc/loss.hip); the torch formulation below is
        # ~35 elementwise launches of a few hundred bytes each, forward and backward
        from . import small_ops
        extras, weights = list(cl.values()), [0.1] * len(cl)
        if "distillation_loss" in outputs:
            extras.append(outputs["distillation_loss"])
            weights.append(0.5)
        return small_ops.fusion_loss(logits, targets, label_smoothing, extras, weights)
    loss = F.cross_entropy(logits, targets, label_smoothing=label_smoothing)
    if cl:
        loss = loss + 0.1 * sum(cl.values())
    if "distillation_loss" in outputs:
        loss = loss + 0.5 * outputs["distillation_loss"]
    return loss


def backward_from(loss: torch.Tensor) -> None:
    """``loss.backward()`` for a loss of ``fusion_loss``: seeds autograd with a resident unit gradient (no fill launch)."""
    if loss.is_cuda:
        from . import small_ops
        small_ops.backward_from(loss)
    else:
        loss.backward()


class FusionTrainStep:
    """fusion module + classifier head + loss + (RCCL all-reduce) + fused AdamW, as one callable.

    ``model(text, audio, video, compute_contrastive_loss=...)`` must return a dict with
    ``fused_features``; ``head`` maps them to emotion logits (reference ``EmotionClassifier``)."""

    def __init__(self, model: torch.nn.Module, head: torch.nn.Module, arena: ParamArena, *, lr: float = 1e-4,
                 weight_decay: float = 1e-5, max_grad_norm: float = 1.0, total_steps: int = 1000,
                 contrastive: bool = True, allreduce: Optional[str] = "bf16", shard_optimizer: bool = False,
                 exchange: str = "after", exchange_rounds: int = 4):
        """exchange: "after" = one bucketed all-reduce of the gradient arena after backward; "backward" = the exchange
        runs inside backward in ``exchange_rounds`` reverse-autograd rounds (``dp.BackwardExchange``; replicated optimiser
        only)."""
        self.model, self.head, self.arena = model, head, arena
        self.opt = FusedAdamW(arena, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, shard=shard_optimizer)
        self.opt.set_schedule(lr, total_steps)            # OneCycleLR evaluated on the device, per step
        self.max_lr, self.total_steps, self.contrastive, self.allreduce = lr, total_steps, contrastive, allreduce
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        if exchange not in ("after", "backward"):
            raise ValueError("exchange must be 'after' or 'backward'")
        self._bx = None
        if exchange == "backward" and self.world > 1:
            if shard_optimizer:
                raise ValueError("the in-backward all-reduce and the sharded optimiser (reduce-scatter) exclude each other")
            self._bx = dp.BackwardExchange(arena, exchange_rounds, None if allreduce == "fp32" else "bf16").install()

    def fwd_bwd(self, text, audio, video, targets) -> torch.Tensor:
        self.arena.zero_grad(overlap=True, lazy=True)
        kw = {"compute_contrastive_loss": True} if self.contrastive else {}
        out = self.model(text, audio, video, **kw)
        fused = out["fused_features"] if isinstance(out, dict) else out
        outputs = dict(out) if isinstance(out, dict) else {}
        outputs["emotion_logits"] = self.head(fused)
        loss = fusion_loss(outputs, targets)
        backward_from(loss)
        self.arena.finalize_grads()
        return loss

    def __call__(self, text, audio, video, targets) -> torch.Tensor:
        loss = self.fwd_bwd(text, audio, video, targets)
        if self._bx is not None:                             # most of the arena is already on the wire
            self._bx.finish()
        elif self.world > 1 and not self.opt.sharded:        # (the sharded step reduce-scatters the gradients itself)
            dp.allreduce_grads(self.arena, compress=None if self.allreduce == "fp32" else "bf16")
        self.opt.advance()
        self.opt.launch()
        return loss
