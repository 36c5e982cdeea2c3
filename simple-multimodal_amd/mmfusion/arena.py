"""Flat parameter / gradient arenas for the fusion path.

All parameters of a fusion module tree are re-homed into ONE contiguous fp32 buffer
(``arena.master``), with

  * ``arena.shadow``  a bf16 copy of the same layout, refreshed by ONE cast launch per step — the
    operand the MFMA GEMMs read (``param._mmf_bf16`` is the per-parameter view);
  * ``arena.grads``   an fp32 buffer of the same layout; ``param.grad`` is a view of it.  The wgrad,
    bias-sum and LayerNorm-backward kernels accumulate straight into it, and it is the single
    buffer RCCL all-reduces under data parallelism (``mmfusion.dp``).  It is pre-zeroed, so
    parameters that receive no gradient in a step (contrastive projectors without the
    contrastive loss, ...) still contribute zeros to the all-reduce instead of ``None``.

``state_dict()`` keys and shapes are untouched (parameters stay ``nn.Parameter`` objects whose
storage is a view of the arena), so reference checkpoints load.
Every parameter starts at a multiple of 64 elements, which keeps fp32, bf16 and row-slice
pointers 16-byte aligned for the kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import lib

ALIGN = 64


class ParamArena:
    def __init__(self, root: torch.nn.Module):
        params: List[torch.nn.Parameter] = []
        seen = set()
        for p in root.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("mmfusion: the fusion path runs on the GPU only; move the module to a "
                               "HIP device first (there is no CPU fallback)")
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("mmfusion: parameters must be fp32 masters")
        self.params = params
        self.offsets: List[int] = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                n = p.numel()
                self.master[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.master[o:o + n].view(p.shape)
                p._mmf_bf16 = self.shadow[o:o + n].view(p.shape)
                p._mmf_arena = self
        self.attach_grads()
        self._cast_version = None
        self.refresh(force=True)

    # -- gradients -----------------------------------------------------------------------------
    def grad_view(self, i: int) -> torch.Tensor:
        p, o = self.params[i], self.offsets[i]
        return self.grads[o:o + p.numel()].view(p.shape)

    def attach_grads(self) -> None:
        """(Re)attach ``param.grad`` views; if any was dropped (``zero_grad(set_to_none=True)``)
        the arena is zeroed, which is exactly what ``None`` grads mean."""
        missing = False
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or g.data_ptr() != self.grads.data_ptr() + 4 * self.offsets[i]:
                missing = True
                break
        if missing:
            self.grads.zero_()
            for i, p in enumerate(self.params):
                p.grad = self.grad_view(i)

    def zero_grad(self) -> None:
        self.grads.zero_()

    # -- bf16 shadow ---------------------------------------------------------------------------
    def _version(self) -> int:
        return sum(p._version for p in self.params)

    def valid(self) -> bool:
        base = self.master.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def mark_shadow_fresh(self) -> None:
        """Called by ``mmfusion.train.FusedAdamW`` after its kernel has written masters AND shadow:
        the next forward needs no cast (until some other writer touches a parameter)."""
        self._cast_version = self._version()
        self._shadow_fresh = True

    def refresh(self, force: bool = False) -> None:
        """master (fp32) -> shadow (bf16): one streaming kernel over the whole arena."""
        v = self._version()
        if getattr(self, "_shadow_fresh", False) and v == self._cast_version:
            return                     # the fused optimiser keeps the shadow in step with the masters
        self._shadow_fresh = False
        if force or v != self._cast_version:
            lib.check(lib.load().mmf_cast_f32_to_bf16(self.master.data_ptr(), self.shadow.data_ptr(),
                                                      self.numel, lib.stream_ptr()))
            self._cast_version = v


def ensure(module: torch.nn.Module, refresh: Optional[bool] = None) -> ParamArena:
    """Make sure every parameter under ``module`` lives in a valid arena, attach gradient views and
    refresh the bf16 shadow.  Cheap when nothing changed.  ``refresh=True`` forces the cast (the
    root call of a training step does that, mirroring autocast's per-forward weight cast);
    otherwise the cast runs only when a parameter's version counter moved."""
    arena: Optional[ParamArena] = None
    ok = True
    for p in module.parameters():
        a = getattr(p, "_mmf_arena", None)
        if a is None or (arena is not None and a is not arena):
            ok = False
            break
        arena = a
    if ok and arena is not None and not arena.valid():
        ok = False
    if not ok or arena is None:
        arena = ParamArena(module)
        module._mmf_arena_root = arena
    arena.attach_grads()
    arena.refresh(force=bool(refresh))
    return arena
