"""fp32-storage parity mode of the fusion path (BASELINE.json north_star: "within 1e-3 fp32").

Selected per root module (``module.precision = "fp32"`` or ``config.fusion_precision = "fp32"`` or
``MMF_PRECISION=fp32``); ``mmfusion.ops`` routes its grouped ops here while such a module's forward runs.  Activations
and weights stay f32 in HBM; every dense contraction runs on the exact f32 MFMA (``csrc/gemmf32.hip``:
v_mfma_f32_32x32x2_f32, a k-ordered fp32 fmaf chain), attention is the reference's explicit-scores form (S = Q K^T,
row softmax, O = P V as batched f32 GEMMs + a softmax kernel pair), LayerNorm is an f32 kernel.  Weight / bias /
LayerNorm gradients go straight into the fp32 gradient arena as on the bf16 path.  What stays on torch ops in this mode
is elementwise / (B, d)-row glue (residual sums, pooling, ReLU masks) — it is a parity instrument: speed is secondary
(fp32 MFMA peak is 1/16 of bf16) and dropout is refused (parity runs use p = 0)."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch

from . import lib
from .lib import (EPI_ACCUM, EPI_ADD_AUX, EPI_BIAS, EPI_COLSUM_A, EPI_MASK_AUX, EPI_RELU, GEMM_NN, GEMM_NT, GEMM_TN,
                  GemmProblem)

F32 = torch.float32


def _ld(t: torch.Tensor) -> int:
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"expected a 2-D row-major (possibly row-strided) tensor, got {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


def _req(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("mmfusion ops run on the GPU only (no CPU fallback)")
    if t.dtype != F32:
        raise TypeError(f"fp32 mode expects float32 tensors, got {t.dtype}")


def gemm_group(layout: int, probs: Sequence[tuple], epilogue: int, alpha: float = 1.0) -> None:
    """probs: (A, B, C, bias|None, aux|None) f32 tensors; M, N from C; K from A."""
    ps: List[GemmProblem] = []
    for (A, Bm, Cm, bias, aux) in probs:
        _req(A), _req(Bm), _req(Cm)
        M, N = Cm.shape
        K = A.shape[0] if layout == GEMM_TN else A.shape[1]
        ok = {GEMM_NT: A.shape == (M, K) and Bm.shape == (N, K), GEMM_NN: A.shape == (M, K) and Bm.shape == (K, N),
              GEMM_TN: A.shape == (K, M) and Bm.shape == (K, N)}[layout]
        if not ok:
            raise ValueError(f"f32 gemm layout {layout}: A{tuple(A.shape)} B{tuple(Bm.shape)} C{tuple(Cm.shape)}")
        if aux is not None and tuple(aux.shape) != (M, N):
            raise ValueError("aux must match C")
        ps.append(GemmProblem(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), bias.data_ptr() if bias is not None else None,
                              aux.data_ptr() if aux is not None else None, M, N, K, _ld(A), _ld(Bm), _ld(Cm),
                              _ld(aux) if aux is not None else 0))
    for i in range(0, len(ps), lib.GEMM_MAX_PROBLEMS):
        chunk = ps[i:i + lib.GEMM_MAX_PROBLEMS]
        arr = (GemmProblem * len(chunk))(*chunk)
        lib.check(lib.load().mmf_gemm_f32_grouped(arr, len(chunk), layout, epilogue, alpha, lib.stream_ptr()))


def gemm(layout, A, Bm, Cm, *, bias=None, aux=None, epilogue=0, alpha=1.0) -> None:
    gemm_group(layout, [(A, Bm, Cm, bias, aux)], epilogue, alpha)


def wgrad(dy: torch.Tensor, x: torch.Tensor, wg: torch.Tensor, bg: Optional[torch.Tensor]) -> None:
    """wg (f32 [N_out, K_in]) += dy^T x ; bg += column sums of dy — issued at once (no deferred queue in this mode);
    honours the arena's lazy zeroing (first touch overwrites)."""
    from .arena import arena_of
    ar = arena_of(wg)
    overwrite = ar.take_first_touch(wg) if ar is not None else False
    gemm(GEMM_TN, dy, x, wg, bias=bg, epilogue=(0 if overwrite else EPI_ACCUM) | (EPI_COLSUM_A if bg is not None else 0))


# --------------------------------------------------------------------------------------------
# grouped Linear
# --------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """tensors = [x_0, res_0|None, w_0.p, b_0.p|None, ...]"""

    @staticmethod
    def forward(ctx, specs, cat: bool, *tensors):
        n = len(specs)
        relu, has_bias, has_res = specs[0].relu, specs[0].b is not None, specs[0].has_residual
        xs, outs, probs = [], [], []
        base, off = None, 0
        if cat:
            base = torch.empty((tensors[0].shape[0], sum(s.w.master.shape[0] for s in specs)), dtype=F32, device=tensors[0].device)
        for i, s in enumerate(specs):
            x, res = tensors[4 * i].float(), tensors[4 * i + 1]
            w = s.w.master
            if cat:
                y = base[:, off:off + w.shape[0]]
                off += w.shape[0]
            else:
                y = torch.empty((x.shape[0], w.shape[0]), dtype=F32, device=x.device)
            probs.append((x, w, y, s.b.master if has_bias else None, res))
            xs.append(x), outs.append(y)
        gemm_group(GEMM_NT, probs, (EPI_BIAS if has_bias else 0) | (EPI_RELU if relu else 0) | (EPI_ADD_AUX if has_res else 0))
        ctx.specs, ctx.cat = specs, cat
        ctx.save_for_backward(*xs, *(([base] if cat else outs) if relu else []))
        ctx.x_needs = [tensors[4 * i].requires_grad for i in range(n)]
        return base if cat else tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        specs = ctx.specs
        n = len(specs)
        xs = ctx.saved_tensors[:n]
        ys = ctx.saved_tensors[n:] if specs[0].relu else None
        dys = []
        if ctx.cat:
            g = gys[0]
            if g is not None:
                g = g.float().contiguous()
                if specs[0].relu:
                    g = g * (ys[0] > 0)
            off = 0
            for s in specs:
                nout = s.w.master.shape[0]
                dys.append(None if g is None else g[:, off:off + nout])
                off += nout
        else:
            for i, g in enumerate(gys):
                if g is None:
                    dys.append(None)
                    continue
                g = g.float().contiguous()
                dys.append(g * (ys[i] > 0) if specs[0].relu else g)
        grads: List[Optional[torch.Tensor]] = [None] * (4 * n)
        dgrad = []
        for i, s in enumerate(specs):
            g = dys[i]
            if g is None:
                continue
            if ctx.x_needs[i]:
                dx = torch.empty(xs[i].shape, dtype=F32, device=g.device)
                dgrad.append((g, s.w.master, dx, None, None))
                grads[4 * i] = dx
            wgrad(g, xs[i], s.w.grad, s.b.grad if s.b is not None else None)
            if s.has_residual:
                grads[4 * i + 1] = g
        if dgrad:
            gemm_group(GEMM_NN, dgrad, 0)
        return (None, None, *grads)


def linear_group(items: Sequence[tuple], cat: bool = False):
    specs, tensors = [], []
    for x, spec, res in items:
        spec.has_residual = res is not None
        specs.append(spec)
        tensors += [x, res, spec.w.p, spec.b.p if spec.b is not None else None]
    out = _Linear.apply(specs, cat, *tensors)
    return out if cat else list(out)


# --------------------------------------------------------------------------------------------
# FFN with residual
# --------------------------------------------------------------------------------------------
class _FFN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layers, *tensors):
        n = len(layers)
        xs = [tensors[5 * i].float().contiguous() for i in range(n)]
        hs = [torch.empty((x.shape[0], l1.weight.shape[0]), dtype=F32, device=x.device) for x, (l1, _) in zip(xs, layers)]
        gemm_group(GEMM_NT, [(x, l1.weight.detach(), h, l1.bias.detach(), None) for x, h, (l1, _) in zip(xs, hs, layers)],
                   EPI_BIAS | EPI_RELU)
        ys = [torch.empty_like(x) for x in xs]
        gemm_group(GEMM_NT, [(h, l2.weight.detach(), y, l2.bias.detach(), x) for x, h, y, (_, l2) in zip(xs, hs, ys, layers)],
                   EPI_BIAS | EPI_ADD_AUX)
        ctx.layers = layers
        ctx.save_for_backward(*xs, *hs)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        layers = ctx.layers
        n = len(layers)
        xs, hs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        idx = [i for i, g in enumerate(gys) if g is not None]
        dys = {i: gys[i].float().contiguous() for i in idx}
        dhs = {i: torch.empty_like(hs[i]) for i in idx}
        dxs = {i: torch.empty_like(xs[i]) for i in idx}
        gemm_group(GEMM_NN, [(dys[i], layers[i][1].weight.detach(), dhs[i], None, hs[i]) for i in idx], EPI_MASK_AUX)
        gemm_group(GEMM_NN, [(dhs[i], layers[i][0].weight.detach(), dxs[i], None, dys[i]) for i in idx], EPI_ADD_AUX)
        for i in idx:
            wgrad(dys[i], hs[i], layers[i][1].weight.grad, layers[i][1].bias.grad)
            wgrad(dhs[i], xs[i], layers[i][0].weight.grad, layers[i][0].bias.grad)
        grads: List[Optional[torch.Tensor]] = [None] * (5 * n)
        for i in idx:
            grads[5 * i] = dxs[i]
        return (None, *grads)


def ffn_residual_group(items: Sequence[tuple]) -> List[torch.Tensor]:
    layers, tensors = [], []
    for x, l1, l2 in items:
        layers.append((l1, l2))
        tensors += [x, l1.weight, l1.bias, l2.weight, l2.bias]
    return list(_FFN.apply(layers, *tensors))


# --------------------------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eps, x, gamma, beta):
        _req(x)
        x = x.contiguous()
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        st = torch.empty((2, rows), dtype=F32, device=x.device)
        lib.check(lib.load().mmf_layernorm_f32_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), st[0].data_ptr(),
                                                   st[1].data_ptr(), rows, d, eps, lib.stream_ptr()))
        ctx.params = (gamma, beta)
        ctx.save_for_backward(x, st)
        return y

    @staticmethod
    def backward(ctx, g):
        x, st = ctx.saved_tensors
        gamma, beta = ctx.params
        g = g.float().contiguous()
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        lib.check(lib.load().mmf_layernorm_f32_bwd(x.data_ptr(), g.data_ptr(), gamma.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                                   dx.data_ptr(), gamma.grad.data_ptr(), beta.grad.data_ptr(), rows, d,
                                                   lib.stream_ptr()))
        return None, dx, None, None


def layernorm_group(items: Sequence[tuple], eps: float = 1e-5) -> List[torch.Tensor]:
    return [_LayerNorm.apply(eps, x, g, b) for x, g, b in items]


# --------------------------------------------------------------------------------------------
# explicit-scores attention (torch F.multi_head_attention_forward, need_weights path)
# --------------------------------------------------------------------------------------------
def _batched(layout, A, a_off, lda, sA, Bm, b_off, ldb, sB, Cm, c_off, ldc, sC, M, N, K, nb0, nb1, alpha=1.0):
    p = GemmProblem(A.data_ptr() + 4 * a_off, Bm.data_ptr() + 4 * b_off, Cm.data_ptr() + 4 * c_off, None, None, M, N, K,
                    lda, ldb, ldc, 0)
    S2 = C.c_int64 * 2
    lib.check(lib.load().mmf_gemm_f32_batched(C.byref(p), layout, 0, alpha, nb0, nb1, S2(*sA), S2(*sB), S2(*sC), lib.stream_ptr()))


class _Attention(torch.autograd.Function):
    """One attention problem (spec) over 2-D f32 source buffers; heads are consecutive head_dim column groups."""

    @staticmethod
    def forward(ctx, spec, H: int, dh: int, *srcs):
        d = H * dh
        B, Tq, Tk = spec.B, spec.Tq, spec.Tk
        qs, ks, vs = srcs[spec.q[0]], srcs[spec.k[0]], srcs[spec.v[0]]
        for t in (qs, ks, vs):
            _req(t)
        dev = qs.device
        P = torch.empty((B, H, Tq, Tk), dtype=F32, device=dev)
        o = torch.empty((B * Tq, d), dtype=F32, device=dev)
        scale = 1.0 / math.sqrt(dh)
        ldq, ldk, ldv = qs.shape[1], ks.shape[1], vs.shape[1]
        # S[b, h] = Q[b, h] K[b, h]^T
        _batched(GEMM_NT, qs, spec.q[1], ldq, (Tq * ldq, dh), ks, spec.k[1], ldk, (Tk * ldk, dh), P, 0, Tk, (H * Tq * Tk, Tq * Tk),
                 Tq, Tk, dh, B, H)
        lib.check(lib.load().mmf_softmax_rows_f32(P.data_ptr(), B * H * Tq, Tk, scale, lib.stream_ptr()))
        # O[b, h] = P[b, h] V[b, h]
        _batched(GEMM_NN, P, 0, Tk, (H * Tq * Tk, Tq * Tk), vs, spec.v[1], ldv, (Tk * ldv, dh), o, 0, d, (Tq * d, dh), Tq, dh, Tk, B, H)
        ctx.spec, ctx.H, ctx.dh, ctx.nsrc = spec, H, dh, len(srcs)
        ctx.save_for_backward(qs, ks, vs, P)
        return o

    @staticmethod
    def backward(ctx, go):
        spec, H, dh = ctx.spec, ctx.H, ctx.dh
        qs, ks, vs, P = ctx.saved_tensors
        d = H * dh
        B, Tq, Tk = spec.B, spec.Tq, spec.Tk
        go = go.float().contiguous()
        scale = 1.0 / math.sqrt(dh)
        ldq, ldk, ldv = qs.shape[1], ks.shape[1], vs.shape[1]
        dev = go.device
        # gradient buffers per distinct source (zero-filled: a source may hold columns this problem does not touch)
        uniq = {}
        for si in (spec.q[0], spec.k[0], spec.v[0]):
            if si not in uniq:
                uniq[si] = torch.zeros_like({spec.q[0]: qs, spec.k[0]: ks, spec.v[0]: vs}[si])
        gq, gk, gv = uniq[spec.q[0]], uniq[spec.k[0]], uniq[spec.v[0]]
        sP = (H * Tq * Tk, Tq * Tk)
        # dV[b, h] = P^T dO
        _batched(GEMM_TN, P, 0, Tk, sP, go, 0, d, (Tq * d, dh), gv, spec.v[1], ldv, (Tk * ldv, dh), Tk, dh, Tq, B, H)
        # dP = dO V^T, then dS = scale * P * (dP - rowsum(dP * P)) in place
        dP = torch.empty_like(P)
        _batched(GEMM_NT, go, 0, d, (Tq * d, dh), vs, spec.v[1], ldv, (Tk * ldv, dh), dP, 0, Tk, sP, Tq, Tk, dh, B, H)
        lib.check(lib.load().mmf_softmax_bwd_rows_f32(P.data_ptr(), dP.data_ptr(), B * H * Tq, Tk, scale, lib.stream_ptr()))
        # dQ = dS K ; dK = dS^T Q
        _batched(GEMM_NN, dP, 0, Tk, sP, ks, spec.k[1], ldk, (Tk * ldk, dh), gq, spec.q[1], ldq, (Tq * ldq, dh), Tq, dh, Tk, B, H)
        _batched(GEMM_TN, dP, 0, Tk, sP, qs, spec.q[1], ldq, (Tq * ldq, dh), gk, spec.k[1], ldk, (Tk * ldk, dh), Tk, dh, Tq, B, H)
        out: List[Optional[torch.Tensor]] = [None] * ctx.nsrc
        for si, g in uniq.items():
            out[si] = g
        return (None, None, None, *out)


def attention_group(specs, H: int, dh: int, srcs: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    return [_Attention.apply(s, H, dh, *srcs) for s in specs]
