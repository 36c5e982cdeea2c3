"""autograd wrappers of the small fused branch kernels (``csrc/small.hip``): dense 3-node GAT layer, L2-normalise
+ symmetric InfoNCE, adaptive softmax-weighted combination, narrow linear heads, node stacking with type
embedding, per-sample modality masks.  Conventions as in ``mmfusion.ops``: GPU only (no fallback), parameter
gradients are accumulated by the kernels straight into ``param.grad`` (the fp32 gradient arena) and the Function
returns ``None`` for them."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib, ops
from .ops import BF16, _req

F32 = torch.float32
NCE_MAX_B = 64
NARROW_MAX_N = 16


def _grad_of(p: torch.nn.Parameter) -> torch.Tensor:
    g = p.grad
    if g is None or g.dtype != F32 or not g.is_contiguous():
        raise RuntimeError("parameter has no fp32 arena gradient: call mmfusion.arena.ensure(module)")
    return g


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------------
# (B, 3d) "cat3" views: the three modality feature matrices side by side
# --------------------------------------------------------------------------------------------
def cat3(t: torch.Tensor, a: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """fp32 (B, 3d) = cat([t, a, v], -1).  When the three are the column thirds of one contiguous fp32 buffer
    (hier-seq: the pooled sequence means) that buffer itself is returned — no copy, gradients flow to it."""
    B, d = t.shape
    base = t._base
    if (base is not None and a._base is base and v._base is base and base.dtype == F32 and base.is_contiguous()
            and tuple(base.shape) == (B, 3 * d) and t.dtype == F32
            and t.data_ptr() == base.data_ptr() and a.data_ptr() == base.data_ptr() + 4 * d
            and v.data_ptr() == base.data_ptr() + 8 * d and t.stride(0) == a.stride(0) == v.stride(0) == 3 * d):
        return base
    return torch.cat([t.float(), a.float(), v.float()], dim=-1)


split3_nocopy_hits = 0          # times _Split3.backward returned the producer's buffer (tests)


class _Split3(torch.autograd.Function):
    """The three column thirds of a (B, 3d) tensor as row-strided views.  Plain slicing would do, but its backward
    is three zero-fills plus three copies; this one is a single ``torch.cat``."""

    @staticmethod
    def forward(ctx, c):
        d = c.shape[1] // 3
        ctx.meta = (c.shape[0], d, c.dtype, c.device)
        return c[:, :d], c[:, d:2 * d], c[:, 2 * d:]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        B, d, dtype, dev = ctx.meta
        # the producer (ops._GroupedLinear.backward) writes the three input gradients as the column thirds of one
        # buffer when its inputs were such thirds: that buffer is the gradient, no concatenation kernel
        if g0 is not None and g1 is not None and g2 is not None:
            base, es = g0._base, g0.element_size()
            if (base is not None and g1._base is base and g2._base is base and base.dtype == dtype and base.is_contiguous()
                    and tuple(base.shape) == (B, 3 * d) and g0.data_ptr() == base.data_ptr()
                    and g1.data_ptr() == base.data_ptr() + es * d and g2.data_ptr() == base.data_ptr() + 2 * es * d
                    and g0.stride(0) == g1.stride(0) == g2.stride(0) == 3 * d):
                global split3_nocopy_hits
                split3_nocopy_hits += 1
                return base
        gs = [g if g is not None else torch.zeros((B, d), dtype=dtype, device=dev) for g in (g0, g1, g2)]
        return torch.cat([g.to(dtype) for g in gs], dim=1)


def split3(c: torch.Tensor):
    return _Split3.apply(c)


class _Stack3Embed(torch.autograd.Function):
    """x[b][m][:] = cat3[b][m*d:(m+1)*d] + emb[m][:]  ->  bf16 (B*3, d) rows (GraphFusion :255-264; emb None =
    plain stacking)."""

    @staticmethod
    def forward(ctx, c3: torch.Tensor, emb: Optional[torch.nn.Parameter]):
        _req(c3, F32)
        c3 = c3.contiguous()
        B, d = c3.shape[0], c3.shape[1] // 3
        x = torch.empty((B * 3, d), dtype=BF16, device=c3.device)
        p = c3.data_ptr()
        lib.check(lib.load().mmf_stack3_embed_fwd(p, p + 4 * d, p + 8 * d, _ptr(emb), x.data_ptr(), B, d, 3 * d,
                                                  lib.stream_ptr()))
        ctx.emb, ctx.B, ctx.d, ctx.need = emb, B, d, c3.requires_grad
        return x

    @staticmethod
    def backward(ctx, dx):
        B, d = ctx.B, ctx.d
        dx = dx.contiguous()
        dc3 = torch.empty((B, 3 * d), dtype=F32, device=dx.device) if ctx.need else None
        p = _ptr(dc3)
        demb = _grad_of(ctx.emb).data_ptr() if ctx.emb is not None else None
        lib.check(lib.load().mmf_stack3_embed_bwd(dx.data_ptr(), p, p + 4 * d if p else None, p + 8 * d if p else None,
                                                  demb, B, d, 3 * d, lib.stream_ptr()))
        return dc3, None


def stack3_embed(c3: torch.Tensor, emb: Optional[torch.nn.Parameter]) -> torch.Tensor:
    return _Stack3Embed.apply(c3, emb)


# --------------------------------------------------------------------------------------------
# dense 3-node GAT layer
# --------------------------------------------------------------------------------------------
class _Gat3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, att_src, att_dst, bias, B: int, heads: int, relu: bool, pool: bool, drop):
        _req(h, F32)
        h = h.contiguous()
        C_ = h.shape[1] // heads
        dev = h.device
        y = torch.empty((B * 3, C_), dtype=BF16, device=dev)
        pooled = torch.empty((B, C_), dtype=BF16, device=dev) if pool else None
        alpha = torch.empty((B, 3, 3, heads), dtype=F32, device=dev)
        sdots = torch.empty((B, 2, 3, heads), dtype=F32, device=dev)
        prm = lib.Gat3Params(B, heads, C_, int(relu), 0.2, drop[0] if drop else 0.0,
                             ops.rng_state().data_ptr() if drop else None, drop[1] if drop else 0)
        lib.check(lib.load().mmf_gat3_dense_fwd(h.data_ptr(), att_src.data_ptr(), att_dst.data_ptr(), bias.data_ptr(),
                                                y.data_ptr(), _ptr(pooled), alpha.data_ptr(), sdots.data_ptr(),
                                                C.byref(prm), lib.stream_ptr()))
        ctx.params, ctx.cfg = (att_src, att_dst, bias), (B, heads, C_, relu, drop)
        ctx.save_for_backward(h, y, alpha, sdots)
        ctx.pool = pool
        if pool:
            return y, pooled
        return y

    @staticmethod
    def backward(ctx, *gs):
        h, y, alpha, sdots = ctx.saved_tensors
        B, heads, C_, relu, drop = ctx.cfg
        att_src, att_dst, bias = ctx.params
        dy = gs[0].contiguous() if gs[0] is not None else None
        dpool = gs[1].contiguous() if ctx.pool and gs[1] is not None else None
        if dy is not None and dy.dtype != BF16:
            dy = ops.cast_to_bf16(dy)
        if dpool is not None and dpool.dtype != BF16:
            dpool = ops.cast_to_bf16(dpool)
        if dy is None and dpool is None:
            return (None,) * 9
        dh = torch.empty_like(h)
        prm = lib.Gat3Params(B, heads, C_, int(relu), 0.2, drop[0] if drop else 0.0,
                             ops.rng_state().data_ptr() if drop else None, drop[1] if drop else 0)
        lib.check(lib.load().mmf_gat3_dense_bwd(h.data_ptr(), att_src.data_ptr(), att_dst.data_ptr(), y.data_ptr(),
                                                alpha.data_ptr(), sdots.data_ptr(), _ptr(dy), _ptr(dpool), dh.data_ptr(),
                                                _grad_of(att_src).data_ptr(), _grad_of(att_dst).data_ptr(),
                                                _grad_of(bias).data_ptr(), C.byref(prm), lib.stream_ptr()))
        return (dh,) + (None,) * 8


def gat3(h: torch.Tensor, att_src, att_dst, bias, B: int, heads: int, relu: bool = True, pool: bool = False,
         dropout_p: float = 0.0):
    """h: f32 (B*3, heads*C) = node features after the layer's linear map.  Returns y bf16 (B*3, C) = relu(GAT(h))
    and, with ``pool``, also the mean over the three nodes, bf16 (B, C)."""
    drop = (float(dropout_p), ops.next_site()) if dropout_p > 0.0 else None
    return _Gat3.apply(h, att_src, att_dst, bias, B, heads, relu, pool, drop)


def gat3_f32(h: torch.Tensor, att_src, att_dst, bias, B: int, heads: int, pool: bool = False):
    """fp32 parity mode: relu(GATConv) on the 3-clique + self loops as f32 torch ops on the (B, 3) nodes (the same
    dense arithmetic as csrc/small.hip gat3_*, reference :267-282 via PyG; parity unpinned).  h: f32 (B*3, heads*C)."""
    Cc = h.shape[1] // heads
    hh = h.view(B, 3, heads, Cc)
    s_src = (hh * att_src.view(1, 1, heads, Cc)).sum(-1)
    s_dst = (hh * att_dst.view(1, 1, heads, Cc)).sum(-1)
    e = torch.nn.functional.leaky_relu(s_dst.unsqueeze(2) + s_src.unsqueeze(1), 0.2)       # (B, i, j, H)
    ex = torch.exp(e - e.max(dim=2, keepdim=True).values)
    alpha = ex / (ex.sum(dim=2, keepdim=True) + 1e-16)
    y = torch.relu(torch.einsum("bijh,bjhc->bihc", alpha, hh).mean(dim=2) + bias)
    rows = y.reshape(B * 3, Cc)
    return (rows, y.mean(dim=1)) if pool else rows


# --------------------------------------------------------------------------------------------
# L2-normalise + symmetric InfoNCE of the three pairs
# --------------------------------------------------------------------------------------------
class _ContrastiveNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, temperature: float, with_loss: bool, z0, z1, z2):
        zs = [z.contiguous() for z in (z0, z1, z2)]
        for z in zs:
            _req(z, F32)
        B, D = zs[0].shape
        dev = zs[0].device
        ns = [torch.empty_like(z) for z in zs]
        inv = torch.empty((3, B), dtype=F32, device=dev)
        losses = torch.empty(3, dtype=F32, device=dev) if with_loss else None
        lse = torch.empty((3, 2, B), dtype=F32, device=dev) if with_loss else None
        P3 = C.c_void_p * 3
        lib.check(lib.load().mmf_infonce_fwd(P3(*[z.data_ptr() for z in zs]), P3(*[n.data_ptr() for n in ns]),
                                             inv.data_ptr(), _ptr(losses), _ptr(lse), B, D, temperature, lib.stream_ptr()))
        ctx.save_for_backward(*ns, inv, *([lse] if with_loss else []))
        ctx.cfg = (B, D, temperature, with_loss)
        if with_loss:
            return (*ns, *losses.unbind(0))
        return tuple(ns)

    @staticmethod
    def backward(ctx, *gs):
        B, D, temperature, with_loss = ctx.cfg
        saved = ctx.saved_tensors
        ns, inv = saved[:3], saved[3]
        lse = saved[4] if with_loss else None
        dn = [g.contiguous() if g is not None else None for g in gs[:3]]
        dl = [g.contiguous().float() if g is not None else None for g in gs[3:6]] if with_loss else [None] * 3
        dz = [torch.empty_like(n) for n in ns]
        P3 = C.c_void_p * 3
        lib.check(lib.load().mmf_infonce_bwd(P3(*[n.data_ptr() for n in ns]), inv.data_ptr(), _ptr(lse),
                                             P3(*[_ptr(g) for g in dn]), P3(*[_ptr(g) for g in dl]),
                                             P3(*[z.data_ptr() for z in dz]), B, D, temperature, lib.stream_ptr()))
        return (None, None, *dz)


def normalize_infonce(zs: Sequence[torch.Tensor], temperature: float, with_loss: bool
                      ) -> Tuple[List[torch.Tensor], Optional[List[torch.Tensor]]]:
    """zs: three f32 (B, D) projections.  Returns the L2-normalised projections and, if ``with_loss``, the
    symmetric InfoNCE losses of the pairs (0,1), (0,2), (1,2) as 0-dim tensors.  B <= 64."""
    out = _ContrastiveNCE.apply(float(temperature), bool(with_loss), *zs)
    return list(out[:3]), (list(out[3:6]) if with_loss else None)


# --------------------------------------------------------------------------------------------
# AdaptiveFusion's weighting
# --------------------------------------------------------------------------------------------
class _AdaptiveCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp, attended, w2, b2):
        _req(hp, F32), _req(attended, F32)
        hp, attended = hp.contiguous(), attended.contiguous()
        B, d = hp.shape
        aw = torch.empty((B, 3), dtype=F32, device=hp.device)
        weighted = torch.empty((B, d), dtype=BF16, device=hp.device)
        lib.check(lib.load().mmf_adaptive_combine_fwd(hp.data_ptr(), w2.data_ptr(), b2.data_ptr(), attended.data_ptr(),
                                                      aw.data_ptr(), weighted.data_ptr(), B, d, lib.stream_ptr()))
        ctx.save_for_backward(hp, attended, aw)
        ctx.params = (w2, b2)
        return weighted, aw

    @staticmethod
    def backward(ctx, dweighted, daw):
        hp, attended, aw = ctx.saved_tensors
        w2, b2 = ctx.params
        B, d = hp.shape
        if dweighted is not None:
            dweighted = dweighted.contiguous()
            if dweighted.dtype != BF16:
                dweighted = ops.cast_to_bf16(dweighted)
        if daw is not None:
            daw = daw.contiguous().float()
        datt = torch.empty_like(attended)
        dhp = torch.empty_like(hp)
        lib.check(lib.load().mmf_adaptive_combine_bwd(hp.data_ptr(), w2.data_ptr(), attended.data_ptr(), aw.data_ptr(),
                                                      _ptr(dweighted), _ptr(daw), datt.data_ptr(), dhp.data_ptr(),
                                                      _grad_of(w2).data_ptr(), _grad_of(b2).data_ptr(), B, d,
                                                      lib.stream_ptr()))
        return dhp, datt, None, None


def adaptive_combine(hp: torch.Tensor, attended: torch.Tensor, w2: torch.nn.Parameter, b2: torch.nn.Parameter):
    """hp f32 (B, d), attended f32 (B, 3, d) -> (weighted bf16 (B, d), adaptive weights f32 (B, 3))."""
    return _AdaptiveCombine.apply(hp, attended, w2, b2)


# --------------------------------------------------------------------------------------------
# narrow linear heads (N <= 16), f32 masters
# --------------------------------------------------------------------------------------------
class _NarrowLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _req(x, F32)
        x = x.contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=F32, device=x.device)
        lib.check(lib.load().mmf_linear_narrow_fwd(x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), M, N, K, lib.stream_ptr()))
        ctx.save_for_backward(x)
        ctx.params, ctx.need = (w, b), x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, b = ctx.params
        M, K = x.shape
        N = w.shape[0]
        dy = dy.contiguous().float()
        dx = torch.empty_like(x) if ctx.need else None
        lib.check(lib.load().mmf_linear_narrow_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dx), _grad_of(w).data_ptr(),
                                                   _grad_of(b).data_ptr() if b is not None else None, M, N, K, lib.stream_ptr()))
        return dx, None, None


def narrow_linear(x: torch.Tensor, layer: torch.nn.Linear) -> torch.Tensor:
    """y = x W^T + b for an nn.Linear with <= 16 outputs (fp32 in, fp32 master weights, fp32 out)."""
    if layer.weight.shape[0] > NARROW_MAX_N:
        raise ValueError("narrow_linear handles at most 16 output features")
    return _NarrowLinear.apply(x.float(), layer.weight, layer.bias)


# --------------------------------------------------------------------------------------------
# per-sample modality masks
# --------------------------------------------------------------------------------------------
class _RowMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        _req(x, F32), _req(mask, F32)
        x, mask = x.contiguous(), mask.contiguous()
        y = torch.empty_like(x)
        B, d = x.shape
        lib.check(lib.load().mmf_rowmask_apply(x.data_ptr(), mask.data_ptr(), y.data_ptr(), B, d, lib.stream_ptr()))
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous().float()
        dx = torch.empty_like(dy)
        B, d = dy.shape
        lib.check(lib.load().mmf_rowmask_apply(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), B, d, lib.stream_ptr()))
        return dx, None


def rowmask(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """y[b, :] = x[b, :] * mask[b]  (fp32 (B, d), fp32 (B,))."""
    return _RowMask.apply(x.float(), mask)


# --------------------------------------------------------------------------------------------
# ModalityDropout and the training-step loss as one launch each (round 4; csrc/loss.hip)
# --------------------------------------------------------------------------------------------
import ctypes as _C


def _ptr3(ts):
    return (_C.c_void_p * 3)(*[t.data_ptr() for t in ts])


class _ModalityDropout(torch.autograd.Function):
    """reference models/encoders.py:289-321 on the three (B, d) f32 feature tensors: masks drawn and applied by ONE kernel
    (mmf_modality_dropout); the backward applies the saved (B, 3) masks with the same kernel."""

    @staticmethod
    def forward(ctx, t, a, v, p, site):
        xs = [x.float().contiguous() for x in (t, a, v)]
        B, d = xs[0].shape
        ys = [torch.empty_like(x) for x in xs]
        keep = torch.empty((B, 3), dtype=F32, device=xs[0].device)
        lib.check(lib.load().mmf_modality_dropout(_ptr3(xs), _ptr3(ys), keep.data_ptr(), B, d, float(p), ops.rng_state().data_ptr(),
                                                  int(site), 1, lib.stream_ptr()))
        ctx.save_for_backward(keep)
        ctx.mark_non_differentiable(keep)
        return ys[0], ys[1], ys[2], keep

    @staticmethod
    def backward(ctx, gt, ga, gv, _gk):
        keep, = ctx.saved_tensors
        B = keep.shape[0]
        ref = next(g for g in (gt, ga, gv) if g is not None)
        gs = [(g if g is not None else torch.zeros_like(ref)).float().contiguous() for g in (gt, ga, gv)]
        d = gs[0].shape[1]
        dx = [torch.empty_like(g) for g in gs]
        lib.check(lib.load().mmf_modality_dropout(_ptr3(gs), _ptr3(dx), keep.data_ptr(), B, d, 0.0, None, 0, 0, lib.stream_ptr()))
        return dx[0], dx[1], dx[2], None, None


def modality_dropout(t: torch.Tensor, a: torch.Tensor, v: torch.Tensor, p: float):
    """-> (t', a', v', keep (B, 3)); masks from the build's counter-based RNG (mmfusion.ops: dropout section)"""
    return _ModalityDropout.apply(t, a, v, float(p), ops.next_site())


_ONE: dict = {}


def loss_seed(device) -> torch.Tensor:
    """the resident d(loss)/d(loss) = 1 tensor of ``backward_from``: a backward started with it costs the fused loss no launch"""
    key = str(device)
    if key not in _ONE:
        _ONE[key] = torch.ones((), dtype=F32, device=device)
    return _ONE[key]


def backward_from(loss: torch.Tensor) -> None:
    """loss.backward() without the fill kernel that builds the unit gradient (and, for ``fusion_loss``, without the
    multiplications by it)"""
    torch.autograd.backward([loss], [loss_seed(loss.device)])


_SCALARS: dict = {}


def _resident_scalar(v: float, device) -> torch.Tensor:
    """a 0-d f32 tensor holding v, created once per (value, device): gradients of constant weight cost no fill launch"""
    key = (v, str(device))
    if key not in _SCALARS:
        _SCALARS[key] = torch.full((), v, dtype=F32, device=device)
    return _SCALARS[key]


class _FusionLoss(torch.autograd.Function):
    """mean CE(label_smoothing) over (B, C) logits + sum_j w_j * extra_j (device scalars): value and d/d(logits) in one launch."""

    @staticmethod
    def forward(ctx, logits, targets, smoothing, weights, *extras):
        logits = logits.float()
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        B, Cn = logits.shape
        ex = [e.float().reshape(1) for e in extras]
        loss = torch.empty((), dtype=F32, device=logits.device)
        dlog = torch.empty((B, Cn), dtype=F32, device=logits.device)
        n = len(ex)
        pe = (_C.c_void_p * max(n, 1))(*[e.data_ptr() for e in ex])
        pw = (_C.c_float * max(n, 1))(*[float(w) for w in weights])
        lib.check(lib.load().mmf_fusion_loss(logits.data_ptr(), logits.stride(0), targets.data_ptr(), B, Cn, float(smoothing),
                                             pe, pw, n, loss.data_ptr(), dlog.data_ptr(), lib.stream_ptr()))
        ctx.save_for_backward(dlog)
        ctx.weights = [float(w) for w in weights]
        ctx.wt = [_resident_scalar(float(w), logits.device) for w in weights]
        return loss

    @staticmethod
    def backward(ctx, g):
        dlog, = ctx.saved_tensors
        if g.data_ptr() == loss_seed(g.device).data_ptr():       # started by backward_from: the gradient IS one
            return (dlog, None, None, None, *ctx.wt)
        return (dlog * g, None, None, None, *[g * w for w in ctx.weights])


def fusion_loss(logits: torch.Tensor, targets: torch.Tensor, smoothing: float, extras, weights) -> torch.Tensor:
    if targets.dtype != torch.int64:
        targets = targets.long()
    return _FusionLoss.apply(logits, targets.contiguous(), float(smoothing), list(weights), *extras)
