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

import os

import torch

from . import lib

ALIGN = 64
SHARD_ALIGN = 64 * 16          # arena storage granularity: equal 64-aligned shards for up to 16 ranks
SMALL_NUMEL = 1 << 16          # 2-D parameters below this size count as "small" (read as fp32 masters by the small kernels)
# Side-stream (fork/join) execution of the late shadow cast and of the gradient zeroing.  OFF by default:
# measured on MI355X (same box, graph replay) it is a wash for MulT (3.09 vs 3.08-3.09 ms: the streams
# compete with the in-projection GEMM for HBM) and a loss for the hierarchical training step (6.30 vs
# 5.71 ms: multi-stream graph replay).  MMF_OVERLAP=1 enables it for A/B runs.
OVERLAP = os.environ.get("MMF_OVERLAP", "0") == "1"


# zero_grad(lazy=True): first-touch overwrite by the wgrad GEMMs instead of a 4 B/param memset (MMF_LAZY_ZERO=0: off)
LAZY_ZERO = os.environ.get("MMF_LAZY_ZERO", "1") != "0"
import weakref
_ARENAS: "weakref.WeakSet[ParamArena]" = weakref.WeakSet()


def arena_of(grad: torch.Tensor) -> Optional["ParamArena"]:
    """The arena whose gradient buffer contains ``grad`` (ops.queue_wgrad: first-touch overwrite)."""
    ptr = grad.data_ptr()
    for a in _ARENAS:
        base = a.grads.data_ptr()
        if base <= ptr < base + 4 * a.numel:
            return a
    return None


def _is_early(name: str) -> bool:
    """Parameters the first launches of a forward read: attention in-projections.  They are laid out at
    the front of the arena so that their bf16 cast can run first while the rest streams on a side stream."""
    return "in_proj" in name


class ParamArena:
    def __init__(self, root: torch.nn.Module):
        named, seen = [], set()
        for n, p in root.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                named.append((n, p))
        if not named:
            raise ValueError("module has no parameters")
        # stable: attention in-projections first, then the other matrices, then the vectors (biases, LayerNorm):
        # the matrices' gradients are produced by the wgrad GEMMs, which can overwrite instead of accumulate
        # (zero_grad(lazy=True)), so what still needs a memset each step is the contiguous tail
        # (round 3) "small" = everything that is not a big 2-D GEMM weight: vectors, (1, H, G) attention vectors, embeddings,
        # the d -> 7 / 3 / 1 heads.  The small kernels read these as fp32 MASTERS (exact, no shadow); the big matrices are
        # only ever read through the bf16 shadow.  Keeping the small ones together at the tail lets the sharded optimiser
        # step refresh their masters on every rank with one tiny collective (FusedAdamW.launch_sharded).
        def _small(q):
            return q.dim() != 2 or q.numel() < SMALL_NUMEL
        named.sort(key=lambda np_: 2 if _small(np_[1]) else (0 if _is_early(np_[0]) else 1))
        params: List[torch.nn.Parameter] = [p for _, p in named]
        n_early = sum(1 for n, q in named if _is_early(n) and not _small(q))
        n_big = sum(1 for _, q in named if not _small(q))
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
        self.early_numel = self.offsets[n_early] if n_early < len(params) else off
        self.small_start = self.offsets[n_big] if n_big < len(params) else off      # first element of the small-parameter tail
        self._side: Optional[torch.cuda.Stream] = None      # fork/join stream for late cast + grad zeroing
        self._pending = False
        # The storage is padded to SHARD_ALIGN elements so that it divides into equal, 64-element-aligned shards for
        # world sizes 2, 4, 8, 16: the sharded optimiser step under data parallelism (mmfusion.train.FusedAdamW(shard=True):
        # in-place reduce-scatter of `grads_full`, all-gather of `shadow_full`) works on the *_full tensors; everything else
        # sees the first `numel` elements.
        self.capacity = (off + SHARD_ALIGN - 1) // SHARD_ALIGN * SHARD_ALIGN
        self.master_full = torch.zeros(self.capacity, dtype=torch.float32, device=dev)
        self.shadow_full = torch.zeros(self.capacity, dtype=torch.bfloat16, device=dev)
        self.grads_full = torch.zeros(self.capacity, dtype=torch.float32, device=dev)
        self.master, self.shadow, self.grads = self.master_full[:off], self.shadow_full[:off], self.grads_full[:off]
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                n = p.numel()
                self.master[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.master[o:o + n].view(p.shape)
                p._mmf_bf16 = self.shadow[o:o + n].view(p.shape)
                p._mmf_arena = self
                p._mmf_late = o >= self.early_numel
        _ARENAS.add(self)
        # gradient regions (element offset -> numel) a wgrad GEMM has produced: whole matrices, or row blocks
        # of one (the q and kv parts of a cross-attention in_proj_weight are written by separate GEMMs)
        self._managed: Dict[int, int] = {}
        self._lazy: Optional[set] = None    # lazy zeroing in force: managed regions not written yet this step
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

    def _fork(self) -> torch.cuda.Stream:
        """Side stream that has waited for everything enqueued on the current stream so far."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.master.device)
        if not self._pending:
            self._side.wait_stream(torch.cuda.current_stream())
            self._pending = True
        return self._side

    def join(self) -> None:
        """Make the current stream wait for the side-stream work (late shadow cast, gradient zeroing).
        Called by the ops before the first use of a late weight; a no-op when nothing is pending."""
        if self._pending:
            torch.cuda.current_stream().wait_stream(self._side)
            self._pending = False

    def zero_grad(self, overlap: bool = False, lazy: bool = False) -> None:
        """overlap=True enqueues the 4 B/param memset on the side stream; it is joined before the first
        late weight is read in the NEXT forward, i.e. long before any backward kernel accumulates.  Only
        for callers that run forward right after (bench.py, mmfusion.train).

        lazy=True (training-step harnesses; the caller MUST call ``finalize_grads()`` after backward and
        before anything reads the gradients): matrices whose gradient the wgrad GEMMs produced in earlier
        steps are not memset — the first wgrad of the step overwrites (``ops.queue_wgrad`` asks
        ``take_first_touch``), later ones accumulate, and ``finalize_grads`` zeroes any that received
        nothing.  Everything else (biases, LayerNorm, torch-produced gradients) is zeroed here as a few
        contiguous ranges.  Saves the 4 B/param memset and the wgrad epilogue's read of the old value."""
        if lazy and LAZY_ZERO and self._managed and self.grads.is_cuda:
            self.join()
            ranges = self._unmanaged_ranges()
            if 0 < len(ranges) <= lib.ZERO_MAX_RANGES:          # every hole in ONE launch
                import ctypes as C
                n = len(ranges)
                st, en = (C.c_int64 * n)(*[r[0] for r in ranges]), (C.c_int64 * n)(*[r[1] for r in ranges])
                lib.check(lib.load().mmf_zero_ranges_f32(self.grads.data_ptr(), st, en, n, lib.stream_ptr()))
            else:
                for s0, e0 in ranges:
                    self.grads[s0:e0].zero_()
            self._lazy = set(self._managed)
            return
        self._lazy = None
        if overlap and OVERLAP and self.grads.is_cuda:
            with torch.cuda.stream(self._fork()):
                self.grads.zero_()
        else:
            self.join()
            self.grads.zero_()

    def _unmanaged_ranges(self) -> List[tuple]:
        """Complement of the managed regions in [0, numel), as coalesced (start, end) element ranges."""
        cached = getattr(self, "_ranges_cache", None)
        if cached is not None and cached[0] == len(self._managed):
            return cached[1]
        ranges, pos = [], 0
        for o in sorted(self._managed):
            if o > pos:
                ranges.append((pos, o))
            pos = max(pos, o + self._managed[o])
        if pos < self.numel:
            ranges.append((pos, self.numel))
        # alignment gaps between parameters hold no gradient: drop ranges that are nothing but padding
        ends = sorted(o + p.numel() for o, p in zip(self.offsets, self.params))
        starts = sorted(self.offsets)
        import bisect

        def only_padding(s0, e0):
            i = bisect.bisect_right(starts, s0) - 1          # parameter containing or preceding s0
            return i >= 0 and s0 >= ends[i] and (i + 1 >= len(starts) or e0 <= starts[i + 1])
        ranges = [r for r in ranges if not only_padding(*r)]
        self._ranges_cache = (len(self._managed), ranges)
        return ranges

    def take_first_touch(self, grad: torch.Tensor) -> bool:
        """Called by ``ops.queue_wgrad`` with the gradient view a wgrad GEMM is about to produce (a whole
        parameter or a contiguous row block of one): records the region as wgrad-managed and returns True
        if the GEMM must OVERWRITE (lazy zeroing in force and nothing has written the region yet this
        step), False if it must accumulate."""
        off = (grad.data_ptr() - self.grads.data_ptr()) // 4
        n = grad.numel()
        if off < 0 or off + n > self.numel or not grad.is_contiguous():
            return False
        known = self._managed.get(off)
        if known is None:
            # a new region must not overlap a managed one (then it keeps accumulating onto memset zeros)
            for o, m in self._managed.items():
                if o < off + n and off < o + m:
                    if self._lazy is not None and o in self._lazy:     # stale values underneath: zero them now
                        self.grads[o:o + m].zero_()
                        self._lazy.discard(o)
                    return False
            if self._lazy is None:
                self._managed[off] = n
            return False
        if known != n:
            return False
        if self._lazy is not None and off in self._lazy:
            self._lazy.discard(off)
            return True
        return False

    def finalize_grads(self) -> None:
        """End of a lazily-zeroed backward: managed regions no wgrad wrote this step hold stale values."""
        from . import ops as _ops
        if self.grads.is_cuda:                       # kernels on the branch streams write gradients too
            _ops.join_branch_streams()
        if self._lazy:
            for off in sorted(self._lazy):
                self.grads[off:off + self._managed[off]].zero_()
        self._lazy = None

    # -- bf16 shadow ---------------------------------------------------------------------------
    def _version(self) -> int:
        return sum(p._version for p in self.params)

    def valid(self) -> bool:
        base = self.master.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    masters_stale = None               # set by FusedAdamW.launch_sharded (ZeRO-1) to the optimiser, cleared by its gather_masters()

    def require_fresh_masters(self, what: str) -> None:
        """A sharded optimiser step leaves this rank's fp32 masters outside its shard one step behind (the bf16 shadow every rank
        computes with IS current).  Anything that reads the masters calls this first: it raises — gathering is a collective
        that cannot be started from one rank's read — and names the fix (silently recasting the shadow from stale
        masters would revert weights on this rank)."""
        if self.masters_stale is not None:
            raise RuntimeError(f"{what}: the fp32 masters of this rank are stale outside its optimiser shard (ZeRO-1 step); call "
                               "FusedAdamW.gather_masters() on every rank first (save_checkpoint does)")

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
            self.require_fresh_masters("recasting the bf16 shadow from the fp32 masters")
            L, e = lib.load(), self.early_numel
            if OVERLAP and 0 < e < self.numel:
                # late block (everything but the in-projections) on the side stream, in parallel with the
                # first GEMMs / attention of the forward; early block on the current stream
                with torch.cuda.stream(self._fork()):
                    lib.check(L.mmf_cast_f32_to_bf16(self.master.data_ptr() + 4 * e, self.shadow.data_ptr() + 2 * e,
                                                     self.numel - e, lib.stream_ptr()))
                lib.check(L.mmf_cast_f32_to_bf16(self.master.data_ptr(), self.shadow.data_ptr(), e, lib.stream_ptr()))
            else:
                lib.check(L.mmf_cast_f32_to_bf16(self.master.data_ptr(), self.shadow.data_ptr(),
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
        if not getattr(module, "_mmf_state_dict_guard", False):
            # module.state_dict() reads the fp32 masters: refuse while a ZeRO-1 step has left them stale on this rank
            def _guard(mod, prefix, keep_vars):
                ar = getattr(mod, "_mmf_arena_root", None)
                if ar is not None:
                    ar.require_fresh_masters("module.state_dict()")
            try:
                module.register_state_dict_pre_hook(_guard)
                module._mmf_state_dict_guard = True
            except AttributeError:                                 # (torch < 2.0)
                pass
    arena.attach_grads()
    arena.refresh(force=bool(refresh))
    return arena
