"""Autograd wrappers around the HIP kernels of ``libmmfusion.so``.

Data layout: activations are bf16, row-major 2-D ``(rows, features)`` with
``rows = B*T``; parameters are fp32 masters living in a flat arena (``mmfusion.arena``) with a
bf16 shadow copy used by the MFMA kernels; weight/bias/LayerNorm gradients are accumulated by
the kernels **directly into ``param.grad``** (an fp32 view of the flat gradient arena, the buffer
RCCL all-reduces), so the Functions return ``None`` for parameters.

Every op is *grouped*: it takes a list of independent problems and issues one launch for all of
them (the six cross-modal blocks of MulT, reference models/fusion_layers.py:146-153, are mutually
independent), which is what fills 256 CUs with these small per-block shapes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib
from .lib import (AttnProblem, GemmProblem, LnProblem, EPI_ACCUM, EPI_ADD_AUX, EPI_BIAS, EPI_COLSUM_A,
                  EPI_DROPOUT, EPI_MASK_AUX, EPI_RELU, GEMM_NN, GEMM_NT, GEMM_TN)

BF16 = torch.bfloat16

# --------------------------------------------------------------------------------------------
# precision mode of the running root forward: "bf16" (default: bf16 storage, bf16 MFMA, f32 accumulate) or "fp32"
# (parity mode: f32 storage, exact f32 MFMA — mmfusion.ops_f32).  Set by models.fusion_layers._FusionBase for the
# duration of a root module's forward; the autograd Functions are separate per mode, so backward needs no flag.
# --------------------------------------------------------------------------------------------
_PRECISION = "bf16"


def set_precision(mode: str) -> str:
    global _PRECISION
    if mode not in ("bf16", "fp32"):
        raise ValueError(f"precision must be 'bf16' or 'fp32', got {mode!r}")
    old, _PRECISION = _PRECISION, mode
    return old


def fp32_mode() -> bool:
    return _PRECISION == "fp32"


def _f32():
    from . import ops_f32
    return ops_f32


# --------------------------------------------------------------------------------------------
# parameter references
# --------------------------------------------------------------------------------------------
@dataclass
class W:
    """Rows [r0, r1) of a (out, in) weight parameter: fp32 master + bf16 shadow."""
    p: torch.nn.Parameter
    r0: int = 0
    r1: Optional[int] = None

    def rows(self) -> Tuple[int, int]:
        return self.r0, (self.p.shape[0] if self.r1 is None else self.r1)

    @property
    def w16(self) -> torch.Tensor:
        return shadow(self.p)[self.rows()[0]:self.rows()[1]]

    @property
    def grad(self) -> torch.Tensor:
        g = self.p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous():
            raise RuntimeError("parameter has no fp32 arena gradient: call mmfusion.arena.ensure(module)")
        a, b = self.rows()
        return g[a:b]

    @property
    def master(self) -> torch.Tensor:
        a, b = self.rows()
        return self.p.detach()[a:b]


def shadow(p: torch.nn.Parameter) -> torch.Tensor:
    """bf16 shadow of an arena-managed parameter.  Parameters outside the arena's early block are cast
    on a side stream: reading one first joins that stream (mmfusion.arena.ParamArena.join)."""
    s = getattr(p, "_mmf_bf16", None)
    if s is None:
        raise RuntimeError("parameter is not arena-managed: call mmfusion.arena.ensure(module) first")
    if _PRECISION == "fp32":
        return p.detach()
    if p._mmf_late:
        p._mmf_arena.join()
    return s


def _ld(t: torch.Tensor) -> int:
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"expected a 2-D row-major (possibly row-strided) tensor, got {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


def _req(t: torch.Tensor, dtype) -> None:
    if not t.is_cuda:
        raise RuntimeError("mmfusion ops run on the GPU only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")


# --------------------------------------------------------------------------------------------
# raw (non-autograd) helpers
# --------------------------------------------------------------------------------------------
def cast_to_bf16(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float32)
    if (x.dim() == 2 and not x.is_contiguous() and x.stride(1) == 1 and x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0):
        # a column block of a wider tensor (what autograd returns for a torch.cat input): cast it where it lies
        y = torch.empty(x.shape, dtype=BF16, device=x.device)
        lib.check(lib.load().mmf_cast_f32_to_bf16_2d(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
                                                     lib.stream_ptr()))
        return y
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    lib.check(lib.load().mmf_cast_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), lib.stream_ptr()))
    return y


def cast_to_f32(x: torch.Tensor) -> torch.Tensor:
    _req(x, BF16)
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    lib.check(lib.load().mmf_cast_bf16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), lib.stream_ptr()))
    return y


def gemm(layout: int, A: torch.Tensor, Bm: torch.Tensor, C: torch.Tensor, *, bias=None, aux=None,
         epilogue: int = 0) -> None:
    """Single-problem convenience over gemm_group (shapes are taken from C and the layout)."""
    gemm_group(layout, [(A, Bm, C, bias, aux)], epilogue)


def gemm_group(layout: int, probs: Sequence[tuple], epilogue: int, alpha: float = 1.0,
               dropout: Optional[tuple] = None) -> None:
    """probs: (A, B, C, bias|None, aux|None) tensors; M,N from C; K from A.
    alpha scales the result after the mask step; dropout = (p, site) adds MMF_EPI_DROPOUT."""
    out_f32 = probs[0][2].dtype == torch.float32
    ps: List[GemmProblem] = []
    for (A, Bm, Cm, bias, aux) in probs:
        _req(A, BF16), _req(Bm, BF16)
        M, N = Cm.shape
        K = A.shape[0] if layout == GEMM_TN else A.shape[1]
        if layout == GEMM_NT:
            ok = A.shape == (M, K) and Bm.shape == (N, K)
        elif layout == GEMM_NN:
            ok = A.shape == (M, K) and Bm.shape == (K, N)
        else:
            ok = A.shape == (K, M) and Bm.shape == (K, N)
        if not ok or (Cm.dtype == torch.float32) != out_f32:
            raise ValueError(f"gemm layout {layout}: A{tuple(A.shape)} B{tuple(Bm.shape)} C{tuple(Cm.shape)}")
        if aux is not None and tuple(aux.shape) != (M, N):
            raise ValueError("aux must match C")
        if bias is not None:
            _req(bias, torch.float32)
            if bias.numel() != (M if epilogue & EPI_COLSUM_A else N) or not bias.is_contiguous():
                raise ValueError("bias must be contiguous [N] ([M] for the fused bias gradient)")
        ps.append(GemmProblem(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(),
                              bias.data_ptr() if bias is not None else None,
                              aux.data_ptr() if aux is not None else None,
                              M, N, K, _ld(A), _ld(Bm), _ld(Cm), _ld(aux) if aux is not None else 0))
    if dropout is not None:
        if len(ps) > lib.GEMM_MAX_PROBLEMS:
            raise ValueError("a dropout GEMM group must fit one launch (the problem index keys the mask)")
        lib.gemm_grouped(ps, layout, epilogue | EPI_DROPOUT, out_f32, alpha, dropout[0],
                         rng_state().data_ptr(), dropout[1])
        return
    for i in range(0, len(ps), lib.GEMM_MAX_PROBLEMS):
        lib.gemm_grouped(ps[i:i + lib.GEMM_MAX_PROBLEMS], layout, epilogue, out_f32, alpha)


# --------------------------------------------------------------------------------------------
# dropout: device-resident RNG state + per-call-site ids
# --------------------------------------------------------------------------------------------
# Masks are a stateless hash of (state, site, element).  `state` is ONE int64 on the device: the root
# forward of a training step adds 1 to it with a (graph-capturable) kernel, so every step — also every
# replay of a captured step — draws new masks, and the backward pass regenerates the forward's masks
# from the same (state, site).  `site` numbers the dropout call sites of one forward in call order.
_rng_state: Optional[torch.Tensor] = None
_site = 0


def rng_state() -> torch.Tensor:
    global _rng_state
    if _rng_state is None:
        _rng_state = torch.full((1,), (torch.initial_seed() & 0x7FFFFFFF) << 20, dtype=torch.int64, device="cuda")
    return _rng_state


def seed_dropout(seed: int) -> None:
    rng_state().fill_((seed & 0x7FFFFFFF) << 20)


def begin_training_forward() -> None:
    """Called once per root forward in training mode: new masks, site numbering restarts."""
    global _site
    _site = 0
    rng_state().add_(1)


def next_site() -> int:
    global _site
    _site += 1
    return _site


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, site):
        x = x.contiguous()
        y = torch.empty_like(x)
        lib.check(lib.load().mmf_dropout(x.data_ptr(), y.data_ptr(), x.numel(), int(x.dtype == torch.float32), p,
                                         rng_state().data_ptr(), site, lib.stream_ptr()))
        ctx.p, ctx.site = p, site
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = torch.empty_like(g)
        lib.check(lib.load().mmf_dropout(g.data_ptr(), dx.data_ptr(), g.numel(), int(g.dtype == torch.float32), ctx.p,
                                         rng_state().data_ptr(), ctx.site, lib.stream_ptr()))
        return dx, None, None


def dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """nn.Dropout(p) on a bf16 or fp32 tensor (identity when not training or p == 0)."""
    if not training or p <= 0.0:
        return x
    if _PRECISION == "fp32":
        raise RuntimeError("the fp32 parity mode runs without dropout (use p = 0 or eval())")
    if x.dtype not in (BF16, torch.float32):
        raise TypeError("dropout expects bf16 or fp32")
    return _Dropout.apply(x, float(p), next_site())


# --------------------------------------------------------------------------------------------
# deferred weight gradients
# --------------------------------------------------------------------------------------------
# dW = dy^T x has a tiny output (768 x 768 ... 3072 x 768) and a long reduction (B*T rows): one
# block's wgrads are ~100 tiles, far fewer than 256 CUs.  Nothing downstream in the backward pass
# reads dW, so every Function only *queues* its wgrad problems and one autograd end-of-backward
# callback issues them all as grouped launches (>= 1000 tiles for MulT), bias gradients included.
import os as _os

DEFER_WGRAD = _os.environ.get("MMF_DEFER_WGRAD", "1") != "0"
_SKINNY = _os.environ.get("MMF_SKINNY", "1") != "0"          # skinny-M kernels for problems with <= 64 rows
_pending_wgrad: List[tuple] = []
# Side-stream wgrad (MMF_WGRAD_SIDE=1, experimental A/B): instead of one flush at the end of backward, the queued
# wgrad problems are issued on a second stream whenever MMF_WGRAD_CHUNK output tiles have accumulated, so the long
# wgrad GEMMs fill the CUs the dgrad / attention / LayerNorm chain leaves idle (launch tails, latency-bound kernels).
_WGRAD_SIDE = _os.environ.get("MMF_WGRAD_SIDE", "0") == "1"
_WGRAD_CHUNK = int(_os.environ.get("MMF_WGRAD_CHUNK", "800"))
_wgrad_stream: Optional[torch.cuda.Stream] = None
# Early flush (MMF_WGRAD_EARLY=1; measured +0.4 % and left OFF by default — with it the dominant kernel shares the chip with
# the input-gradient sums, so its duration in a profile of the timed steps no longer equals its stand-alone duration, which
# is what bench.py's roofline is built on): the number of weight-gradient problems a backward queues is the same every
# step, so when the count of the previous backward is reached the whole deferred launch is issued at once on its own
# stream instead of from the end-of-backward callback.  What the backward still has to do after its last Linear (the
# N-way input-gradient sums of the fan-outs: three HBM-bound kernels, 40 us + launch gaps at the MulT bench shapes)
# then runs beside the wgrad launch instead of in front of it.  A backward that queues more, or fewer, than the last one
# is still complete: whatever is pending when the callback runs is issued there, and the callback joins the stream.
_WGRAD_EARLY = _os.environ.get("MMF_WGRAD_EARLY", "0") == "1"
_expected_wgrad = 0
_early_issued = 0
_callback_queued = False
_pending_tiles = 0


_branch_streams: List[torch.cuda.Stream] = []


def branch_stream(i: int = 0) -> torch.cuda.Stream:
    """Extra stream i for work that runs beside the main chain (HierarchicalFusion's small branches, the concurrent
    halves of MulT's cross blocks).  The deferred wgrad flush and anything else that consumes its results on the
    main stream joins every branch stream first (``join_branch_streams``)."""
    while len(_branch_streams) <= i:
        _branch_streams.append(torch.cuda.Stream())
    return _branch_streams[i]


def join_branch_streams() -> None:
    cur = torch.cuda.current_stream()
    for s in _branch_streams:
        cur.wait_stream(s)


_WGRAD_SORT = _os.environ.get("MMF_WGRAD_SORT", "k")


def _wgrad_rounds(pend: List[tuple]) -> List[List[tuple]]:
    """Split queued problems into launch rounds such that no two problems of one round write overlapping gradient
    memory (a weight used twice in one forward, e.g. a shared ``nn.Linear``): the GEMM epilogue's accumulate is a
    plain read-add-write, so two writers of one region inside ONE grouped launch would race and drop updates.  The
    k-th writer of a region goes to round k; queue order is kept inside a region, so its first-touch overwrite
    (lazy zeroing) still runs before its accumulates."""
    rounds: List[List[tuple]] = []
    spans: List[Tuple[int, int, int]] = []               # (start, end, round) of every destination placed so far
    def span(t: torch.Tensor) -> Tuple[int, int]:
        # bytes the kernel may touch: for a row-strided view (a column block of a packed weight) the last row ends at
        # (rows - 1) * ld + cols, not at numel
        if t.dim() == 2:
            return t.data_ptr(), t.data_ptr() + 4 * ((t.shape[0] - 1) * t.stride(0) + t.shape[1])
        return t.data_ptr(), t.data_ptr() + 4 * t.numel()

    for q in pend:
        mine = [span(q[2])]
        if q[3] is not None:
            mine.append(span(q[3]))
        r = 0
        for a, b in mine:
            for s0, e0, r0 in spans:
                if a < e0 and s0 < b:
                    r = max(r, r0 + 1)
        for a, b in mine:
            spans.append((a, b, r))
        while len(rounds) <= r:
            rounds.append([])
        rounds[r].append(q)
    return rounds


def _issue_wgrad(pend: List[tuple]) -> None:
    for rnd in _wgrad_rounds(pend):
        # (has bias, overwrite) -> problems; overwrite = first wgrad of a lazily-zeroed step (arena.zero_grad(lazy=True))
        groups: dict = {}
        for q in rnd:
            groups.setdefault((q[3] is not None, q[5]), []).append(q[:5])
        # longest tiles first (a tile's duration goes with K = the rows of dy), big outputs first among equals: the
        # kernel hands tiles to the CUs in this order, so the tail of the launch is made of the short tiles.
        # Overwrite groups before accumulate groups.
        for (has_bias, overwrite), group in sorted(groups.items(), key=lambda kv: (not kv[0][1], not kv[0][0])):
            if _WGRAD_SORT == "k":
                group.sort(key=lambda t: (-t[0].shape[0], -(t[2].shape[0] * t[2].shape[1])))
            else:
                group.sort(key=lambda t: -(t[2].shape[0] * t[2].shape[1] * t[0].shape[0]))
            gemm_group(GEMM_TN, group, (0 if overwrite else EPI_ACCUM) | (EPI_COLSUM_A if has_bias else 0))


def _issue_wgrad_side(pend: List[tuple], all_streams: bool = False) -> None:
    global _wgrad_stream
    if _wgrad_stream is None:
        _wgrad_stream = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    _wgrad_stream.wait_stream(cur)                       # the queued dy / x are complete on the main stream
    if all_streams:                                      # ... and on the branch streams other backward nodes ran on
        for st in _branch_streams:
            if st != cur:
                _wgrad_stream.wait_stream(st)
    with torch.cuda.stream(_wgrad_stream):
        _issue_wgrad(pend)
    for q in pend:                                       # keep the operands' memory until the side kernels ran
        q[0].record_stream(_wgrad_stream)
        q[1].record_stream(_wgrad_stream)


# Manual flush (data-parallel overlap, bench.py / mmfusion.dp): the end-of-backward callback only parks the queued
# problems; the caller takes them (take_pending_wgrad), splits them by gradient-arena offset (split_wgrad_by_offset)
# and issues the parts itself (issue_wgrad) with the all-reduce of a finished arena range started in between.
_MANUAL_FLUSH = False
_parked_wgrad: List[tuple] = []


def set_manual_wgrad_flush(on: bool) -> None:
    global _MANUAL_FLUSH
    if not DEFER_WGRAD and on:
        raise RuntimeError("manual wgrad flush needs the deferred wgrad queue (MMF_DEFER_WGRAD=1)")
    _MANUAL_FLUSH = bool(on)


def take_pending_wgrad() -> List[tuple]:
    """The problems the last backward parked (manual flush mode), in queue order; the list is handed over."""
    global _parked_wgrad
    pend, _parked_wgrad = _parked_wgrad, []
    return pend


def wgrad_offset(q: tuple) -> int:
    """Element offset of a queued problem's weight gradient inside its gradient arena."""
    from .arena import arena_of
    ar = arena_of(q[2])
    if ar is None:
        raise ValueError("queued weight gradient does not live in a gradient arena")
    return (q[2].data_ptr() - ar.grads.data_ptr()) // 4


def split_wgrad_by_offset(pend: List[tuple], nparts: int = 2) -> Tuple[List[List[tuple]], List[int]]:
    """Partition queued problems into `nparts` consecutive gradient-arena ranges of about equal GEMM work
    (sum of M N K).  Returns (parts, bounds): part i holds every problem whose gradient starts in
    [bounds[i], bounds[i+1]); bounds[0] = 0 and the last bound is open (the caller's arena size).  Problems that write
    the same region fall in the same part, so the overwrite-then-accumulate order inside a part is kept."""
    items = sorted(((wgrad_offset(q), i, q) for i, q in enumerate(pend)), key=lambda t: (t[0], t[1]))
    work = [q[2].shape[0] * q[2].shape[1] * q[0].shape[0] for _, _, q in items]
    total, acc, cuts = float(sum(work)) or 1.0, 0.0, []
    for k in range(len(items)):
        if len(cuts) < nparts - 1 and k > 0 and items[k][0] != items[k - 1][0] and acc >= total * (len(cuts) + 1) / nparts:
            cuts.append(k)
        acc += work[k]
    edges = [0] + cuts + [len(items)]
    parts = [[q for _, _, q in items[edges[i]:edges[i + 1]]] for i in range(len(edges) - 1)]
    bounds = [0] + [items[c][0] for c in cuts]
    while len(parts) < nparts:                            # fewer distinct regions than parts: empty trailing parts
        parts.append([])
        bounds.append(bounds[-1] if not items else items[-1][0] + items[-1][2][2].numel())
    return parts, bounds


def issue_wgrad(pend: List[tuple]) -> None:
    """Issue queued wgrad problems now, on the current stream (grouped launches, as the automatic flush does)."""
    if pend:
        _issue_wgrad(pend)


# Round flush (data-parallel exchange INSIDE backward, SURVEY 8e / VERDICT r2 item 5a): with a hook installed, the backward
# hands its queued weight-gradient problems over in `n` rounds in reverse-autograd order — round i as soon as i/n of the
# problems the PREVIOUS backward queued have been queued again — and whatever is left from the end-of-backward callback
# (final=True).  The hook issues a round's GEMMs and starts the all-reduce of the gradient regions they complete
# (mmfusion.dp.BackwardExchange), so the exchange of the early rounds runs beside the rest of backward.  A region with
# several writers in one backward (a weight used twice) only completes with its last writer: such problems are held back
# for the final round (the set is learned from the previous backward, like the count).  The first backward after
# installing the hook has nothing to learn from and delivers everything in the final round.
_ROUND_HOOK = None
_ROUND_N = 0
_round_issued = 0
_multi_writer: set = set()
_round_ptrs: List[int] = []          # gradient pointers handed over in this backward's earlier rounds
_rounds_cut = 0                      # rounds handed over so far in this backward


def set_wgrad_rounds(n: int, hook) -> None:
    """hook(problems, final) or None to remove.  Needs the deferred queue (MMF_DEFER_WGRAD=1); excludes the manual flush."""
    global _ROUND_HOOK, _ROUND_N, _round_issued, _expected_wgrad
    if hook is not None and (not DEFER_WGRAD or _MANUAL_FLUSH or n < 1):
        raise RuntimeError("wgrad rounds need the deferred wgrad queue, no manual flush, n >= 1")
    _ROUND_HOOK, _ROUND_N, _round_issued = hook, int(n), 0
    _expected_wgrad = 0


def _maybe_cut_round() -> None:
    """queue_wgrad: hand over a round when its share of the expected problems has been queued."""
    global _pending_wgrad, _round_issued
    if _ROUND_HOOK is None or not _expected_wgrad or _ROUND_N < 2:
        return
    global _rounds_cut
    done = _round_issued + len(_pending_wgrad)
    k = _rounds_cut + 1                                                # the round being filled, 1-based
    if k >= _ROUND_N or done * _ROUND_N < k * _expected_wgrad:
        return                                                         # (the last round is the end-of-backward callback's)
    _rounds_cut = k
    ready = [q for q in _pending_wgrad if q[2].data_ptr() not in _multi_writer]
    if not ready:
        return
    _pending_wgrad = [q for q in _pending_wgrad if q[2].data_ptr() in _multi_writer]
    _round_issued += len(ready)
    _round_ptrs.extend(q[2].data_ptr() for q in ready)
    _ROUND_HOOK(ready, False)


def _flush_wgrad() -> None:
    global _pending_wgrad, _callback_queued, _pending_tiles, _expected_wgrad, _early_issued, _round_issued, _multi_writer
    pend, _pending_wgrad = _pending_wgrad, []
    _callback_queued, _pending_tiles = False, 0
    early, _early_issued = _early_issued, 0
    global _rounds_cut
    rounds_done, _round_issued, _rounds_cut = _round_issued, 0, 0
    _expected_wgrad = early + rounds_done + len(pend)    # what the next backward is expected to queue
    if _ROUND_HOOK is not None:
        seen: dict = {}
        for ptr in _round_ptrs + [q[2].data_ptr() for q in pend]:
            seen[ptr] = seen.get(ptr, 0) + 1
        _multi_writer = {ptr for ptr, c in seen.items() if c > 1}
        _round_ptrs.clear()
        _ROUND_HOOK(pend, True)
        return
    if _branch_streams:                                  # operands queued by backward nodes that ran on a branch stream
        cur = torch.cuda.current_stream()
        join_branch_streams()
        for q in pend:
            q[0].record_stream(cur)
            q[1].record_stream(cur)
    if _MANUAL_FLUSH:
        _parked_wgrad.extend(pend)
        return
    if _WGRAD_SIDE:
        if pend:
            _issue_wgrad_side(pend)
        if _wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(_wgrad_stream)      # gradients complete when backward returns
        return
    if early and _wgrad_stream is not None:
        # gradients complete when backward returns — and BEFORE any left-over problem is issued here: a left-over may
        # accumulate into a region the early launch is still writing (a weight used twice)
        torch.cuda.current_stream().wait_stream(_wgrad_stream)
    if pend:
        _issue_wgrad(pend)


def queue_wgrad(dy: torch.Tensor, x: torch.Tensor, wgrad: torch.Tensor, bgrad: Optional[torch.Tensor]) -> None:
    """wgrad (f32, [N_out, K_in]) += dy^T x ;  bgrad (f32 [N_out]) += column sums of dy.
    If the gradient lives in an arena that was zeroed lazily this step and nothing has written it yet, the
    GEMM overwrites instead (``ParamArena.take_first_touch``)."""
    global _callback_queued, _pending_tiles, _pending_wgrad
    overwrite = False
    from .arena import arena_of
    ar = arena_of(wgrad)
    if ar is not None:
        overwrite = ar.take_first_touch(wgrad)
    if not DEFER_WGRAD:
        gemm_group(GEMM_TN, [(dy, x, wgrad, bgrad, None)],
                   (0 if overwrite else EPI_ACCUM) | (EPI_COLSUM_A if bgrad is not None else 0))
        return
    if not _callback_queued:
        torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrad)
        _callback_queued = True
    _pending_wgrad.append((dy, x, wgrad, bgrad, None, overwrite))
    _maybe_cut_round()
    global _early_issued
    if (_WGRAD_EARLY and not _WGRAD_SIDE and not _MANUAL_FLUSH and _ROUND_HOOK is None and _expected_wgrad and not _early_issued
            and len(_pending_wgrad) == _expected_wgrad):
        pend, _pending_wgrad = _pending_wgrad, []
        _early_issued = len(pend)
        _issue_wgrad_side(pend, all_streams=True)
    if _WGRAD_SIDE:
        _pending_tiles += ((wgrad.shape[0] + 255) // 256) * ((wgrad.shape[1] + 127) // 128)
        if _pending_tiles >= _WGRAD_CHUNK:
            pend, _pending_wgrad = _pending_wgrad, []
            _pending_tiles = 0
            _issue_wgrad_side(pend)


# --------------------------------------------------------------------------------------------
# dtype boundary
# --------------------------------------------------------------------------------------------
class _ToBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return cast_to_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return cast_to_f32(g)


class _ToF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return cast_to_f32(x)

    @staticmethod
    def backward(ctx, g):
        return cast_to_bf16(g)


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """To the activation dtype of the running mode: bf16 (f32 in the fp32 parity mode)."""
    if _PRECISION == "fp32":
        return x if x.dtype == torch.float32 else _ToF32.apply(x)
    return x if x.dtype == BF16 else _ToBF16.apply(x)


def to_f32(x: torch.Tensor) -> torch.Tensor:
    return x if x.dtype == torch.float32 else _ToF32.apply(x)


# --------------------------------------------------------------------------------------------
# grouped Linear:  y_i = act(x_i W_i^T + b_i) (+ residual_i)
# --------------------------------------------------------------------------------------------
@dataclass
class LinearSpec:
    w: W
    b: Optional[W] = None          # bias parameter rows [r0, r1) (1-D parameter)
    relu: bool = False
    has_residual: bool = False


def _relu_mask(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dz (bf16) = g * (y > 0) for a ReLU output y; g and y may be f32 (an f32-output linear) — narrowed and masked
    by ONE kernel instead of cast, cast, mask (these are (B, d)-row tensors: a launch costs more than the work)."""
    g, y = g.contiguous(), y.contiguous()
    if g.shape != y.shape:
        raise ValueError("relu mask: gradient and saved output differ in shape")
    dz = torch.empty(g.shape, dtype=BF16, device=g.device)
    L, st = lib.load(), lib.stream_ptr()
    if g.dtype == BF16 and y.dtype == BF16:
        lib.check(L.mmf_relu_bwd_bf16(g.data_ptr(), y.data_ptr(), dz.data_ptr(), g.numel(), st))
    else:
        for t in (g, y):
            if t.dtype not in (BF16, torch.float32):
                raise TypeError("relu mask: bf16 or f32 tensors only")
        lib.check(L.mmf_relu_bwd_mixed(g.data_ptr(), int(g.dtype == torch.float32), y.data_ptr(),
                                       int(y.dtype == torch.float32), dz.data_ptr(), g.numel(), st))
    return dz


class _GroupedLinear(torch.autograd.Function):
    """tensors = [x_0, res_0|None, w_0.p, b_0.p|None, x_1, ...]; returns one output per spec."""

    @staticmethod
    def forward(ctx, specs: List[LinearSpec], out_f32, *tensors):
        # out_f32 may be (out_f32, cat): with cat the outputs are the column blocks of ONE (M, sum N_i) tensor,
        # which is returned instead of the list (torch.cat / torch.stack of the outputs without a copy, and one
        # gradient tensor coming back instead of n slices)
        cat = False
        if isinstance(out_f32, tuple):
            out_f32, cat = out_f32
        n = len(specs)
        relu = specs[0].relu
        has_bias = specs[0].b is not None
        has_res = specs[0].has_residual
        if any(s.relu != relu or (s.b is not None) != has_bias or s.has_residual != has_res for s in specs):
            raise ValueError("a linear group must share one epilogue")
        xs, ress, outs, probs = [], [], [], []
        base, off = None, 0
        if cat:
            M0 = tensors[0].shape[0]
            if any(tensors[4 * i].shape[0] != M0 for i in range(n)):
                raise ValueError("a concatenated linear group needs equal row counts")
            base = torch.empty((M0, sum(s.w.w16.shape[0] for s in specs)), dtype=torch.float32 if out_f32 else BF16,
                               device=tensors[0].device)
        f32_in = []
        for i, s in enumerate(specs):
            x, res = tensors[4 * i], tensors[4 * i + 1]
            # an f32 input (a module's f32 output fed on, e.g. MulT's pooled projections) is narrowed here and its
            # gradient written as f32 by the dgrad kernel: no bf16 -> f32 cast node in the backward chain
            f32_in.append(x.dtype == torch.float32)
            if f32_in[-1]:
                x = cast_to_bf16(x.contiguous())
            _req(x, BF16)
            w16 = s.w.w16
            bias = s.b.master if has_bias else None
            if cat:
                y = base[:, off:off + w16.shape[0]]
                off += w16.shape[0]
            else:
                y = torch.empty((x.shape[0], w16.shape[0]), dtype=torch.float32 if out_f32 else BF16, device=x.device)
            probs.append((x, w16, y, bias, res))
            xs.append(x), ress.append(res), outs.append(y)
        epi = (EPI_BIAS if has_bias else 0) | (EPI_RELU if relu else 0) | (EPI_ADD_AUX if has_res else 0)
        skinny = _SKINNY and not has_res and all(x.shape[0] <= lib.SKINNY_MAX_M and x.shape[1] % 8 == 0 for x in xs)
        if skinny:          # (B, d)-row problems: weight-streaming kernels instead of the 256-row tile
            lib.skinny_fwd([lib.SkinnyProblem(x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                              b.data_ptr() if b is not None else None, None,
                                              x.shape[0], w.shape[0], x.shape[1], _ld(x), _ld(w), _ld(y), 0)
                            for (x, w, y, b, _) in probs], epi, out_f32)
        else:
            gemm_group(GEMM_NT, probs, epi)
        if any(f32_in) and not all(f32_in):
            raise ValueError("a linear group's inputs must share one dtype")
        ctx.f32_in = bool(f32_in and f32_in[0])
        ctx.specs, ctx.out_f32, ctx.cat = specs, out_f32, cat
        ctx.save_for_backward(*xs, *(([base] if cat else outs) if relu else []))
        ctx.x_needs = [tensors[4 * i].requires_grad for i in range(n)]
        return base if cat else tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        specs = ctx.specs
        n = len(specs)
        saved = ctx.saved_tensors
        xs = saved[:n]
        ys = saved[n:] if specs[0].relu else None
        L = lib.load()
        st = lib.stream_ptr()
        dys = []
        if ctx.cat:                                # one gradient for the concatenated output: slice it, no copies
            g = gys[0]
            if g is not None:
                if specs[0].relu:
                    g = _relu_mask(g, ys[0])
                else:
                    g = g.contiguous()
                    if g.dtype != BF16:
                        g = cast_to_bf16(g)
            off = 0
            for s in specs:
                nout = s.w.w16.shape[0]
                dys.append(None if g is None else g[:, off:off + nout])
                off += nout
            gys = dys
        else:
          for i, g in enumerate(gys):
            if g is None:
                dys.append(None)
                continue
            if specs[0].relu:                      # dz = dy * (y > 0)
                g = _relu_mask(g, ys[i])
            else:
                g = cast_to_bf16(g) if g.dtype != BF16 else g.contiguous()      # (the cast takes row-strided sources)
            dys.append(g)
        dgrad = []
        has_bias = specs[0].b is not None
        grads: List[Optional[torch.Tensor]] = [None] * (4 * n)
        # inputs that are the adjacent column blocks of one contiguous tensor (small_ops.split3 of a pooled (B, 3d)
        # buffer) get their gradients as the same blocks of one buffer, so the split's backward has nothing to copy
        dbase, col = None, 0
        x0 = xs[0]
        if (n > 1 and not ctx.f32_in and all(ctx.x_needs) and all(g is not None for g in dys) and x0._base is not None
                and x0._base.is_contiguous() and x0._base.dim() == 2 and x0.data_ptr() == x0._base.data_ptr()
                and x0._base.shape[1] == sum(x.shape[1] for x in xs) and x0._base.shape[0] == x0.shape[0]):
            off, ok = 0, True
            for x in xs:
                ok = ok and x._base is x0._base and x.stride(0) == x0._base.shape[1] and \
                    x.data_ptr() == x0._base.data_ptr() + 2 * off and x.shape[0] == x0.shape[0]
                off += x.shape[1]
            if ok:
                dbase = torch.empty(x0._base.shape, dtype=BF16, device=x0.device)
        for i, s in enumerate(specs):
            g = dys[i]
            if g is None:
                continue
            if ctx.x_needs[i]:
                if dbase is not None:
                    dx = dbase[:, col:col + xs[i].shape[1]]
                    col += xs[i].shape[1]
                else:
                    dx = torch.empty(xs[i].shape, dtype=torch.float32 if ctx.f32_in else BF16, device=g.device)
                dgrad.append((g, s.w.w16, dx, None, None))
                grads[4 * i] = dx
            # wgrad; the bias gradient (column sums of dy) rides along in the same kernel
            queue_wgrad(g, xs[i], s.w.grad, s.b.grad if has_bias else None)
            if s.has_residual:
                grads[4 * i + 1] = g
        if dgrad:
            if _SKINNY and all(g.shape[0] <= lib.SKINNY_MAX_M for (g, _, _, _, _) in dgrad):
                lib.skinny_dgrad([lib.SkinnyProblem(g.data_ptr(), w.data_ptr(), dx.data_ptr(), None, None,
                                                    g.shape[0], w.shape[0], w.shape[1], _ld(g), _ld(w), _ld(dx), 0)
                                  for (g, w, dx, _, _) in dgrad], 0, 1.0, ctx.f32_in)
            else:
                gemm_group(GEMM_NN, dgrad, 0)
        return (None, None, *grads)


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """2-D, last dimension contiguous (row-strided views — column blocks of a wider tensor — pass as they are)"""
    if t.dim() != 2:
        raise ValueError(f"expected a 2-D tensor, got {tuple(t.shape)}")
    return t if t.stride(1) == 1 else t.contiguous()


class _RowLinear(torch.autograd.Function):
    """(B, d)-row linears — M <= 64 rows: the Early / Contrastive / Adaptive / Graph / meta / classifier MLPs and the encoder
    projection tails — with the launches AROUND them folded in (round 4; mmf_skinny_linear_*_ex).  At these sizes a launch costs
    ~5 us and the arithmetic nothing: a Linear + ReLU + Dropout fed by an f32 tensor was cast, linear, dropout (3 launches)
    forward and dropout, ReLU mask, (cast), dgrad (3-4) backward; here it is one launch each way:
      forward   y = dropout(act(x W^T + b)): x may be f32 (narrowed while loaded; the bf16 copy the weight gradient needs is
                written by the kernel's first workgroup), the dropout mask is drawn in the epilogue, and with ``dual`` a second
                copy of y in the other dtype comes out of the same launch (the bf16 operand of the next linear beside the f32
                value the module returns);
      backward  dz = g * (y > 0) / (1 - p) — a ReLU + dropout output is exactly 0 where either gate is closed — or, for dropout
                without ReLU, the regenerated mask, is formed while the dgrad kernel loads g (bf16 or f32, row-strided views
                welcome); dz (bf16) is written once for the deferred weight-gradient launch.
    tensors = [x_0, w_0.p, b_0.p | None, x_1, ...]; returns the n outputs (+ n second copies with ``dual``)."""

    @staticmethod
    def forward(ctx, specs: List[LinearSpec], opts, *tensors):
        out_f32, p, dual = opts
        n = len(specs)
        relu, has_bias = specs[0].relu, specs[0].b is not None
        if any(s.relu != relu or (s.b is not None) != has_bias or s.has_residual for s in specs):
            raise ValueError("a row-linear group must share one epilogue and has no residual form")
        xs = [_rows2d(tensors[3 * i]) for i in range(n)]
        f32_in = xs[0].dtype == torch.float32
        for x in xs:
            if not x.is_cuda:
                raise RuntimeError("mmfusion ops run on the GPU only (no CPU fallback)")
            if x.dtype not in (BF16, torch.float32) or (x.dtype == torch.float32) != f32_in:
                raise TypeError("a row-linear group's inputs must share one dtype (bf16 or f32)")
        site = next_site() if p > 0.0 else 0
        probs, outs, outs2, x16s = [], [], [], []
        for i, s in enumerate(specs):
            x, w16 = xs[i], s.w.w16
            M, K, N = x.shape[0], x.shape[1], w16.shape[0]
            y = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=x.device)
            y2 = torch.empty((M, N), dtype=BF16 if out_f32 else torch.float32, device=x.device) if dual else None
            x16 = torch.empty((M, K), dtype=BF16, device=x.device) if f32_in else x
            bias = s.b.master if has_bias else None
            probs.append(lib.SkinnyProblemEx(
                lib.SkinnyProblem(x.data_ptr(), w16.data_ptr(), y.data_ptr(), bias.data_ptr() if bias is not None else None, None,
                                  M, N, K, _ld(x), _ld(w16), _ld(y), 0),
                y2.data_ptr() if y2 is not None else None, None, x16.data_ptr() if f32_in else None,
                _ld(y2) if y2 is not None else 0, 0, _ld(x16) if f32_in else 0, 0))
            outs.append(y), outs2.append(y2), x16s.append(x16)
        flags = (EPI_BIAS if has_bias else 0) | (EPI_RELU if relu else 0) | (EPI_DROPOUT if p > 0.0 else 0)
        extra = lib.SkinnyExtra(int(f32_in), 0, 1.0, float(p), rng_state().data_ptr() if p > 0.0 else None, site, 0)
        lib.skinny_fwd_ex(probs, flags, out_f32, extra)
        ctx.specs, ctx.opts, ctx.site, ctx.f32_in = specs, opts, site, f32_in
        ctx.save_for_backward(*x16s, *(outs if relu else []))
        ctx.x_needs = [tensors[3 * i].requires_grad for i in range(n)]
        return tuple(outs) + (tuple(outs2) if dual else ())

    @staticmethod
    def backward(ctx, *gs):
        specs = ctx.specs
        out_f32, p, dual = ctx.opts
        n = len(specs)
        relu, has_bias = specs[0].relu, specs[0].b is not None
        saved = ctx.saved_tensors
        x16s, ys = saved[:n], (saved[n:] if relu else [None] * n)
        grads: List[Optional[torch.Tensor]] = [None] * (3 * n)
        todo = []
        for i in range(n):
            g, g2 = gs[i], (gs[n + i] if dual else None)
            if g is not None and g2 is not None:             # both copies of the output were used downstream
                g = g.float() + g2.float()
            elif g is None:
                g = g2
            if g is not None:
                g = _rows2d(g)
                if g.data_ptr() % 16 or g.stride(0) % (4 if g.dtype == torch.float32 else 8):
                    g = g.contiguous()                       # (a view whose rows lose the 16-byte alignment of the vector loads)
                todo.append((i, g))
        # one launch per gradient dtype (the kernel narrows f32 gradients itself)
        for want_f32 in (False, True):
            grp = [(i, g) for i, g in todo if (g.dtype == torch.float32) == want_f32]
            if not grp:
                continue
            gated = relu or p > 0.0
            probs, dxs = [], []
            for i, g in grp:
                w16 = specs[i].w.w16
                M, N, K = g.shape[0], w16.shape[0], w16.shape[1]
                if g.dtype not in (BF16, torch.float32):
                    raise TypeError("row-linear backward: bf16 or f32 gradients only")
                dz = torch.empty((M, N), dtype=BF16, device=g.device) if (gated or want_f32) else g
                dx = torch.empty((M, K), dtype=torch.float32 if ctx.f32_in else BF16, device=g.device)
                y = ys[i]
                probs.append(lib.SkinnyProblemEx(
                    lib.SkinnyProblem(g.data_ptr(), w16.data_ptr(), dx.data_ptr(), None, None, M, N, K, _ld(g), _ld(w16), _ld(dx), 0),
                    None, y.data_ptr() if y is not None else None, dz.data_ptr() if dz is not g else None,
                    0, _ld(y) if y is not None else 0, _ld(dz) if dz is not g else 0, 0))
                dxs.append(dx)
                queue_wgrad(dz, x16s[i], specs[i].w.grad, specs[i].b.grad if has_bias else None)
                if ctx.x_needs[i]:
                    grads[3 * i] = dx
            regen = p > 0.0 and not relu                      # dropout without ReLU: the mask is drawn again from (state, site)
            extra = lib.SkinnyExtra(int(want_f32), int(relu and ys[grp[0][0]].dtype == torch.float32),
                                    1.0 / (1.0 - p) if (relu and p > 0.0) else 1.0, float(p) if regen else 0.0,
                                    rng_state().data_ptr() if regen else None, ctx.site, 0)
            lib.skinny_dgrad_ex(probs, EPI_DROPOUT if regen else 0, 1.0, ctx.f32_in, extra)
        return (None, None, *grads)


def row_linear_ok(items: Sequence[tuple]) -> bool:
    """whether a linear group can take the fused (B, d)-row form: bf16 mode, <= 64 rows, no residual, one launch"""
    if _PRECISION == "fp32" or not _SKINNY or len(items) > lib.SKINNY_MAX_PROBLEMS:
        return False
    for x, spec, res in items:
        if res is not None or x.dim() != 2 or x.shape[0] > lib.SKINNY_MAX_M or x.shape[1] % 8 or spec.w.w16.shape[0] % 4:
            return False
        if x.dtype == torch.float32 and (x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16):
            return False
        if x.dtype == BF16 and (x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16):
            return False
    return True


def linear_group(items: Sequence[tuple], out_f32: bool = False, cat: bool = False, dropout_p: float = 0.0, dual: bool = False):
    """items: (x_bf16 [M,K], LinearSpec, residual_bf16|None).  One NT launch for the group.  Returns the list of
    outputs, or with ``cat`` ONE (M, sum N_i) tensor whose column blocks are the outputs (equal M required)."""
    """``dropout_p``: nn.Dropout(p) on every output (training-mode probability: pass 0 in eval).  ``dual``: every output also as
    a second tensor in the OTHER dtype — the return value is then a list of (y, y_other) pairs.  Groups of (B, d)-row problems
    (``row_linear_ok``) run both inside the linear's own launch (``_RowLinear``); otherwise they are separate kernels."""
    if _PRECISION == "fp32":
        ys = _f32().linear_group(items, cat)
        if dropout_p > 0.0:
            raise RuntimeError("the fp32 parity mode runs without dropout (use p = 0 or eval())")
        return [(y, y) for y in ys] if dual else ys
    if not cat and row_linear_ok(items):
        specs, tensors = [], []
        for x, spec, res in items:
            spec.has_residual = False
            specs.append(spec)
            tensors += [x, spec.w.p, spec.b.p if spec.b is not None else None]
        outs = _RowLinear.apply(specs, (bool(out_f32), float(dropout_p), bool(dual)), *tensors)
        n = len(specs)
        return [(outs[i], outs[n + i]) for i in range(n)] if dual else list(outs)
    specs, tensors = [], []
    for x, spec, res in items:
        spec.has_residual = res is not None
        specs.append(spec)
        tensors += [x, res, spec.w.p, spec.b.p if spec.b is not None else None]
    if cat:
        ys = _GroupedLinear.apply(specs, (out_f32, True), *tensors)
        if dropout_p > 0.0 or dual:
            raise ValueError("a concatenated linear group has no dropout / dual form")
        return ys
    ys = [dropout(y, dropout_p, True) for y in _GroupedLinear.apply(specs, out_f32, *tensors)]
    if dual:
        return [(y, (to_bf16(y) if out_f32 else to_f32(y))) for y in ys]
    return ys


def linear(x: torch.Tensor, w: W, b: Optional[W] = None, relu: bool = False,
           residual: Optional[torch.Tensor] = None, out_f32: bool = False, dropout_p: float = 0.0, dual: bool = False):
    return linear_group([(x, LinearSpec(w, b, relu), residual)], out_f32, dropout_p=dropout_p, dual=dual)[0]


# --------------------------------------------------------------------------------------------
# grouped position-wise FFN with residual:  y_i = x_i + W2_i relu(W1_i x_i + b1_i) + b2_i
# (reference models/fusion_layers.py:195-200,208-209 before norm2).  Fused so that the backward can
# apply the ReLU mask and the residual-gradient add in the dgrad GEMM epilogues:
#   dH = (dY W2) * (h > 0)        NN GEMM, MASK_AUX epilogue (aux = h)
#   dX = dH W1 + dY               NN GEMM, ADD_AUX epilogue  (aux = dY)
# --------------------------------------------------------------------------------------------
class _GroupedFFN(torch.autograd.Function):
    """tensors = [x_0, W1_0, b1_0, W2_0, b2_0, x_1, ...] (parameters passed so autograd runs backward)."""

    @staticmethod
    def forward(ctx, layers, drop, *tensors):
        n = len(layers)
        xs = [tensors[5 * i] for i in range(n)]
        for x in xs:
            _req(x, BF16)
        hs = [torch.empty((x.shape[0], l1.weight.shape[0]), dtype=BF16, device=x.device) for x, (l1, _) in zip(xs, layers)]
        # h = dropout(relu(x W1^T + b1)) in one epilogue; dropped units are exactly 0 in h
        gemm_group(GEMM_NT, [(x, shadow(l1.weight), h, l1.bias.detach(), None)
                             for x, h, (l1, _) in zip(xs, hs, layers)], EPI_BIAS | EPI_RELU, dropout=drop)
        ctx.keep_scale = 1.0 / (1.0 - drop[0]) if drop is not None else 1.0
        ys = [torch.empty_like(x) for x in xs]
        gemm_group(GEMM_NT, [(h, shadow(l2.weight), y, l2.bias.detach(), x)
                             for x, h, y, (_, l2) in zip(xs, hs, ys, layers)], EPI_BIAS | EPI_ADD_AUX)
        ctx.layers = layers
        ctx.save_for_backward(*xs, *hs)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        layers = ctx.layers
        n = len(layers)
        xs, hs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        idx = [i for i, g in enumerate(gys) if g is not None]
        dys = {i: gys[i].contiguous() for i in idx}
        dhs = {i: torch.empty_like(hs[i]) for i in idx}
        dxs = {i: torch.empty_like(xs[i]) for i in idx}
        # (h > 0) is ReLU mask AND dropout mask at once; only the 1/(1-p) factor is left to apply
        gemm_group(GEMM_NN, [(dys[i], shadow(layers[i][1].weight), dhs[i], None, hs[i]) for i in idx], EPI_MASK_AUX,
                   alpha=ctx.keep_scale)
        gemm_group(GEMM_NN, [(dhs[i], shadow(layers[i][0].weight), dxs[i], None, dys[i]) for i in idx], EPI_ADD_AUX)
        for i in idx:
            queue_wgrad(dys[i], hs[i], layers[i][1].weight.grad, layers[i][1].bias.grad)
            queue_wgrad(dhs[i], xs[i], layers[i][0].weight.grad, layers[i][0].bias.grad)
        grads: List[Optional[torch.Tensor]] = [None] * (5 * n)
        for i in idx:
            grads[5 * i] = dxs[i]
        return (None, None, *grads)


def ffn_residual_group(items: Sequence[tuple], dropout_p: float = 0.0) -> List[torch.Tensor]:
    """items: (x_bf16 [M, d], linear1 (d -> 4d), linear2 (4d -> d)); returns x + ffn(x) per item.
    dropout_p > 0: nn.Dropout on the hidden activations (reference :198), fused into the first GEMM."""
    if _PRECISION == "fp32":
        if dropout_p > 0.0:
            raise RuntimeError("the fp32 parity mode runs without dropout (use p = 0 or eval())")
        return _f32().ffn_residual_group(items)
    layers, tensors = [], []
    for x, l1, l2 in items:
        for p in (l1.weight, l1.bias, l2.weight, l2.bias):
            if getattr(p, "_mmf_bf16", None) is None or p.grad is None:
                raise RuntimeError("FFN parameters are not arena-managed: call mmfusion.arena.ensure(module)")
        layers.append((l1, l2))
        tensors += [x, l1.weight, l1.bias, l2.weight, l2.bias]
    drop = (float(dropout_p), next_site()) if dropout_p > 0.0 else None
    return list(_GroupedFFN.apply(layers, drop, *tensors))


# --------------------------------------------------------------------------------------------
# grouped LayerNorm
# --------------------------------------------------------------------------------------------
class _GroupedLayerNorm(torch.autograd.Function):
    """tensors = [x_0, gamma_0, beta_0, x_1, ...]"""

    @staticmethod
    def forward(ctx, eps: float, *tensors):
        n = len(tensors) // 3
        d = tensors[0].shape[-1]
        outs, stats, probs, keep = [], [], [], []
        for i in range(n):
            x, g, b = tensors[3 * i:3 * i + 3]
            _req(x, BF16)
            x = x.contiguous()
            keep.append(x)
            rows = x.numel() // d
            y = torch.empty_like(x)
            st = torch.empty((2, rows), dtype=torch.float32, device=x.device)
            probs.append(LnProblem(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), st[0].data_ptr(),
                                   st[1].data_ptr(), None, None, None, None, rows))
            outs.append(y), stats.append(st)
        for i in range(0, n, lib.LN_MAX_PROBLEMS):
            lib.layernorm_fwd_grouped(probs[i:i + lib.LN_MAX_PROBLEMS], d, eps)
        ctx.n, ctx.d = n, d
        ctx.params = [(tensors[3 * i + 1], tensors[3 * i + 2]) for i in range(n)]
        ctx.save_for_backward(*[tensors[3 * i] for i in range(n)], *stats)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        n, d = ctx.n, ctx.d
        xs, stats = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        grads: List[Optional[torch.Tensor]] = [None] * (3 * n)
        probs, keep = [], []
        for i, g in enumerate(gys):
            if g is None:
                continue
            g = g.contiguous()
            x = xs[i].contiguous()
            keep += [g, x]
            gamma, beta = ctx.params[i]
            if gamma.grad is None or beta.grad is None:
                raise RuntimeError("LayerNorm parameters have no arena gradient")
            dx = torch.empty_like(x)
            rows = x.numel() // d
            probs.append(LnProblem(x.data_ptr(), None, gamma.data_ptr(), None, stats[i][0].data_ptr(),
                                   stats[i][1].data_ptr(), g.data_ptr(), dx.data_ptr(),
                                   gamma.grad.data_ptr(), beta.grad.data_ptr(), rows))
            grads[3 * i] = dx
        ws_bytes = lib.load().mmf_layernorm_bwd_workspace_bytes(d)
        for i in range(0, len(probs), lib.LN_MAX_PROBLEMS):
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=xs[0].device)
            keep.append(ws)
            lib.layernorm_bwd_grouped(probs[i:i + lib.LN_MAX_PROBLEMS], d, ws)
        return (None, *grads)


def layernorm_group(items: Sequence[tuple], eps: float = 1e-5) -> List[torch.Tensor]:
    """items: (x_bf16 [..., d], gamma_param, beta_param)."""
    if _PRECISION == "fp32":
        return _f32().layernorm_group(items, eps)
    flat = []
    for x, g, b in items:
        flat += [x, g, b]
    return list(_GroupedLayerNorm.apply(eps, *flat))


# --------------------------------------------------------------------------------------------
# grouped fused attention
# --------------------------------------------------------------------------------------------
@dataclass
class AttnSpec:
    """One attention problem.  q/k/v are (source index, first column) into the `srcs` list of 2-D
    bf16 buffers (rows = B*T, row-major); heads are laid out as consecutive head_dim column groups."""
    B: int
    Tq: int
    Tk: int
    q: Tuple[int, int]
    k: Tuple[int, int]
    v: Tuple[int, int]


class _GroupedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, specs: List[AttnSpec], H: int, dh: int, drop, *srcs):
        scale = 1.0 / math.sqrt(dh)
        d = H * dh
        outs, lses, probs = [], [], []
        for s in specs:
            for (si, col), T in ((s.q, s.Tq), (s.k, s.Tk), (s.v, s.Tk)):
                t = srcs[si]
                _req(t, BF16)
                if t.dim() != 2 or not t.is_contiguous() or t.shape[0] != s.B * T or col + d > t.shape[1]:
                    raise ValueError(f"attention source {si}: shape {tuple(t.shape)} does not hold B*T={s.B * T} rows x {d} columns at {col}")
            dev = srcs[0].device
            o = torch.empty((s.B * s.Tq, d), dtype=BF16, device=dev)
            lse = torch.empty((s.B * H * s.Tq,), dtype=torch.float32, device=dev)
            qs, ks, vs = srcs[s.q[0]], srcs[s.k[0]], srcs[s.v[0]]
            probs.append(AttnProblem(qs.data_ptr() + 2 * s.q[1], ks.data_ptr() + 2 * s.k[1], vs.data_ptr() + 2 * s.v[1],
                                     o.data_ptr(), lse.data_ptr(), None, None, None, None, None,
                                     s.B, H, s.Tq, s.Tk, qs.shape[1], ks.shape[1], vs.shape[1], d))
            outs.append(o), lses.append(lse)
        if drop is not None and len(probs) > lib.ATTN_MAX_PROBLEMS:
            raise ValueError("an attention group with dropout must fit one launch (the problem index keys the mask)")
        dp_, st_, site_ = (drop[0], rng_state().data_ptr(), drop[1]) if drop is not None else (0.0, None, 0)
        for i in range(0, len(probs), lib.ATTN_MAX_PROBLEMS):
            lib.attn_fwd_grouped(probs[i:i + lib.ATTN_MAX_PROBLEMS], dh, scale, dp_, st_, site_)
        ctx.specs, ctx.H, ctx.dh, ctx.nsrc, ctx.drop = specs, H, dh, len(srcs), drop
        ctx.save_for_backward(*srcs, *outs, *lses)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gos):
        specs, H, dh, ns = ctx.specs, ctx.H, ctx.dh, ctx.nsrc
        n = len(specs)
        d = H * dh
        srcs = ctx.saved_tensors[:ns]
        outs = ctx.saved_tensors[ns:ns + n]
        lses = ctx.saved_tensors[ns + n:]
        # one gradient buffer per source; zero-filled only when some column range is not covered
        covered = [set() for _ in range(ns)]
        for s, g in zip(specs, gos):
            if g is None:
                continue
            for si, col in (s.q, s.k, s.v):
                covered[si].add(col)
        gsrc: List[Optional[torch.Tensor]] = []
        for si, t in enumerate(srcs):
            if not covered[si]:
                gsrc.append(None)
                continue
            full = len(covered[si]) * d == t.shape[1]
            gsrc.append(torch.empty_like(t) if full else torch.zeros_like(t))
        probs, keep = [], []          # `keep`: every temporary whose pointer sits in a problem table must
        for i, (s, g) in enumerate(zip(specs, gos)):      # outlive the launch, or the caching allocator
            if g is None:                                 # hands its block to the next problem
                continue
            g = g.contiguous()
            delta = torch.empty_like(lses[i])
            keep += [g, delta]
            qs, ks, vs = srcs[s.q[0]], srcs[s.k[0]], srcs[s.v[0]]
            probs.append(AttnProblem(qs.data_ptr() + 2 * s.q[1], ks.data_ptr() + 2 * s.k[1], vs.data_ptr() + 2 * s.v[1],
                                     outs[i].data_ptr(), lses[i].data_ptr(), g.data_ptr(), delta.data_ptr(),
                                     gsrc[s.q[0]].data_ptr() + 2 * s.q[1], gsrc[s.k[0]].data_ptr() + 2 * s.k[1],
                                     gsrc[s.v[0]].data_ptr() + 2 * s.v[1],
                                     s.B, H, s.Tq, s.Tk, qs.shape[1], ks.shape[1], vs.shape[1], d))
        drop = ctx.drop
        if drop is not None and len(probs) != len(specs):
            raise RuntimeError("attention dropout backward needs the gradient of every problem of the group")
        dp_, st_, site_ = (drop[0], rng_state().data_ptr(), drop[1]) if drop is not None else (0.0, None, 0)
        for i in range(0, len(probs), lib.ATTN_MAX_PROBLEMS):
            lib.attn_bwd_grouped(probs[i:i + lib.ATTN_MAX_PROBLEMS], dh, 1.0 / math.sqrt(dh), dp_, st_, site_)
        return (None, None, None, None, *gsrc)


def attention_group(specs: List[AttnSpec], H: int, dh: int, srcs: Sequence[torch.Tensor],
                    dropout_p: float = 0.0) -> List[torch.Tensor]:
    """Each (source, column) pair may be the k or v of several problems only if those problems'
    gradients are wanted separately — within one call every (source, column) range must be
    written by at most one problem's dK/dV (true for MulT: each block has its own K/V projection)."""
    if _PRECISION == "fp32":
        if dropout_p > 0.0:
            raise RuntimeError("the fp32 parity mode runs without dropout (use p = 0 or eval())")
        return _f32().attention_group(specs, H, dh, srcs)
    drop = (float(dropout_p), next_site()) if dropout_p > 0.0 else None
    return list(_GroupedAttention.apply(specs, H, dh, drop, *srcs))


# --------------------------------------------------------------------------------------------
# elementwise / pooling
# --------------------------------------------------------------------------------------------
class _Add3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        _req(a, BF16)
        a, b = a.contiguous(), b.contiguous()
        c = c.contiguous() if c is not None else None
        y = torch.empty_like(a)
        lib.check(lib.load().mmf_add3_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr() if c is not None else None,
                                           y.data_ptr(), a.numel(), lib.stream_ptr()))
        ctx.has_c = c is not None
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g, (g if ctx.has_c else None)


def add3(a, b, c=None):
    if _PRECISION == "fp32":
        return a + b if c is None else a + b + c
    return _Add3.apply(a, b, c)


class _Add3Group(torch.autograd.Function):
    """tensors = [a_0, b_0, c_0, a_1, ...] (bf16): y_i = a_i + b_i + c_i for all i in ONE launch."""

    @staticmethod
    def forward(ctx, *tensors):
        n = len(tensors) // 3
        probs, outs, keep = [], [], []
        for i in range(n):
            a, b, c = (t.contiguous() for t in tensors[3 * i:3 * i + 3])
            _req(a, BF16), _req(b, BF16), _req(c, BF16)
            if a.shape != b.shape or a.shape != c.shape:
                raise ValueError("add3 group: operands of one sum must have one shape")
            y = torch.empty_like(a)
            keep += [a, b, c]
            outs.append(y)
            probs.append(lib.Add3Problem(a.data_ptr(), b.data_ptr(), c.data_ptr(), y.data_ptr(), a.numel()))
        arr = (lib.Add3Problem * n)(*probs)
        lib.check(lib.load().mmf_add3_grouped(arr, n, lib.stream_ptr()))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        out = []
        for g in gs:
            out += [g, g, g]
        return tuple(out)


def add3_group(triples: Sequence[tuple]) -> List[torch.Tensor]:
    """[(a, b, c), ...] -> [a + b + c, ...], one launch (up to lib.ADD3_MAX sums)."""
    if _PRECISION == "fp32":
        return [a + b + c for a, b, c in triples]
    if not 0 < len(triples) <= lib.ADD3_MAX:
        raise ValueError(f"add3 group of {len(triples)} sums (1..{lib.ADD3_MAX})")
    return list(_Add3Group.apply(*[t for tr in triples for t in tr]))


class _Fanout(torch.autograd.Function):
    """n aliases of one bf16 tensor for n consumers; the backward receives all n gradients at once and sums them in
    ONE pass (mmf_addn_bf16, f32 accumulate) instead of the n - 1 pairwise adds autograd would issue."""

    @staticmethod
    def forward(ctx, x, n: int):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        live = [g.contiguous() if g.dtype == BF16 else cast_to_bf16(g.contiguous()) for g in live]
        out = torch.empty_like(live[0])
        import ctypes as C
        for i in range(0, len(live), lib.ADDN_MAX - 1):            # chunks of <= 8 operands (running sum carried)
            chunk = live[i:i + lib.ADDN_MAX - 1] + ([out] if i > 0 else [])
            if len(chunk) == 1:
                chunk.append(torch.zeros_like(out))
            ptrs = (C.c_void_p * len(chunk))(*[g.data_ptr() for g in chunk])
            lib.check(lib.load().mmf_addn_bf16(ptrs, len(chunk), out.data_ptr(), out.numel(), 0, lib.stream_ptr()))
        return out, None


def fanout(x: torch.Tensor, n: int) -> List[torch.Tensor]:
    """n handles on x (bf16) whose gradients are summed by one kernel.  Identity when x needs no gradient."""
    if n < 2 or not x.requires_grad or x.dtype != BF16 or _PRECISION == "fp32":
        return [x] * n
    return list(_Fanout.apply(x, n))


class _FanoutGroup(torch.autograd.Function):
    """_Fanout for several tensors at once: n aliases of each of m bf16 tensors; the backward receives all m * n gradients
    together and sums them per tensor in ONE grouped launch (mmf_addn_grouped).  MulT's three modalities' input-gradient
    sums all complete at the end of the backward; as three nodes they were three launches with graph-node gaps between
    them, alone on the chip in front of the deferred wgrad launch (57 us per step in round 3's timeline)."""

    @staticmethod
    def forward(ctx, n: int, *xs):
        ctx.n, ctx.m = n, len(xs)
        return tuple(x.view_as(x) for x in xs for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        n, m = ctx.n, ctx.m
        outs: List[Optional[torch.Tensor]] = []
        probs, keep = [], []
        for i in range(m):
            live = [g for g in gs[i * n:(i + 1) * n] if g is not None]
            if not live:
                outs.append(None)
                continue
            if len(live) == 1:
                outs.append(live[0])
                continue
            live = [g.contiguous() if g.dtype == BF16 else cast_to_bf16(g.contiguous()) for g in live]
            if len(live) > lib.ADDN_MAX or live[0].numel() % 8 or any(g.data_ptr() % 16 for g in live):
                # a gradient the grouped kernel cannot take (a view at an odd offset, an over-wide fan-out): sum this tensor's
                # gradients with stock adds in f32 rather than raising from inside backward
                acc = live[0].float()
                for g in live[1:]:
                    acc = acc + g.float()
                outs.append(acc.to(BF16))
                continue
            out = torch.empty_like(live[0])
            q = lib.AddNProblem()
            for k, g in enumerate(live):
                q.x[k] = g.data_ptr()
            q.y, q.numel, q.n = out.data_ptr(), out.numel(), len(live)
            probs.append(q)
            keep += live
            outs.append(out)
        for i in range(0, len(probs), lib.ADDN_GROUP_MAX):
            chunk = probs[i:i + lib.ADDN_GROUP_MAX]
            arr = (lib.AddNProblem * len(chunk))(*chunk)
            lib.check(lib.load().mmf_addn_grouped(arr, len(chunk), lib.stream_ptr()))
        return (None, *outs)


def fanout_group(xs: Sequence[torch.Tensor], n: int) -> List[List[torch.Tensor]]:
    """[fanout(x, n) for x in xs] whose backward sums every tensor's n gradients in one grouped launch.  Falls back to
    per-tensor fanout when some tensor needs no gradient (or in the fp32 parity mode)."""
    if n < 2 or n > lib.ADDN_MAX or _PRECISION == "fp32" or not all(x.requires_grad and x.dtype == BF16 for x in xs):
        return [fanout(x, n) for x in xs]
    flat = _FanoutGroup.apply(n, *xs)
    return [list(flat[i * n:(i + 1) * n]) for i in range(len(xs))]


class _MeanPoolCat(torch.autograd.Function):
    """xs: (B, T_i, d) bf16 -> (B, n*d) bf16 = cat_i mean_t x_i  (fusion_layers.py:166-171)."""

    @staticmethod
    def forward(ctx, *xs):
        B, _, d = xs[0].shape
        n = len(xs)
        y = torch.empty((B, n * d), dtype=BF16, device=xs[0].device)
        L, st = lib.load(), lib.stream_ptr()
        xs = [x.contiguous() for x in xs]
        for x in xs:
            _req(x, BF16)
        if n <= lib.POOL_MAX:                            # all modalities in one launch
            import ctypes as C
            ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
            Ts = (C.c_int * n)(*[x.shape[1] for x in xs])
            lib.check(L.mmf_meanpool_cat_fwd(ptrs, Ts, n, y.data_ptr(), B, d, n * d, st))
        else:
            for i, x in enumerate(xs):
                lib.check(L.mmf_meanpool_fwd(x.data_ptr(), y.data_ptr() + 2 * i * d, B, x.shape[1], d, n * d, st))
        ctx.shapes = [tuple(x.shape) for x in xs]
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        L, st = lib.load(), lib.stream_ptr()
        n = len(ctx.shapes)
        out = [torch.empty((B, T, d), dtype=BF16, device=g.device) for (B, T, d) in ctx.shapes]
        B, _, d = ctx.shapes[0]
        if n <= lib.POOL_MAX:
            import ctypes as C
            ptrs = (C.c_void_p * n)(*[o.data_ptr() for o in out])
            Ts = (C.c_int * n)(*[sh[1] for sh in ctx.shapes])
            lib.check(L.mmf_meanpool_cat_bwd(g.data_ptr(), ptrs, Ts, n, B, d, n * d, st))
        else:
            for i, (B, T, d) in enumerate(ctx.shapes):
                lib.check(L.mmf_meanpool_bwd(g.data_ptr() + 2 * i * d, out[i].data_ptr(), B, T, d, n * d, st))
        return tuple(out)


def meanpool_cat(xs: Sequence[torch.Tensor]) -> torch.Tensor:
    if _PRECISION == "fp32":
        return torch.cat([x.float().mean(dim=1) for x in xs], dim=-1)
    return _MeanPoolCat.apply(*xs)
