"""Bidirectional multi-layer LSTM on the HIP path (reference models/encoders.py:183-190,233:
``nn.LSTM(768, 384, num_layers=2, batch_first=True, bidirectional=True, dropout=fusion_dropout)`` over the ViT frames).

Per layer (``csrc/lstm.hip`` has the kernel-side description):
  * input projections of all T steps and both directions: ONE grouped MFMA GEMM launch -> gx (T*B, 2*4H) f32;
  * the T sequential steps of both directions: ONE persistent launch, W_hh resident in registers;
  * backward: ONE persistent launch producing the gate pre-activation gradients dG (bf16), then the input gradient
    (two NN GEMMs) and the four weight gradients + four bias gradients through the deferred grouped wgrad launch
    (``ops.queue_wgrad``): dW_ih = dG^T x, dW_hh = dG^T h_prev where h_prev is a shifted VIEW of the layer output.
The parameters stay the ``torch.nn.LSTM`` module's own (``weight_ih_l0``, ``weight_hh_l0_reverse``, ...: the reference's
state_dict keys), arena-managed like every other parameter of the path.  Layout inside is time-major (T, B, .).
GPU only; there is no fallback (``torch.nn.LSTM.forward`` is never called on this path)."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

from . import lib, ops
from .lib import EPI_ADD_AUX, GEMM_NN, GEMM_NT
from .ops import BF16, W, _req

F32 = torch.float32
MAX_B = 64


class _Swap01(torch.autograd.Function):
    """(n0, n1, d) -> (n1, n0, d) as bf16 rows (the (B, T, d) <-> (T, B, d) change of major axis), f32 or bf16 in."""

    @staticmethod
    def forward(ctx, x, out_f32):
        if not x.is_cuda:
            raise RuntimeError("mmfusion LSTM runs on the GPU only (no CPU fallback)")
        if x.dtype not in (BF16, F32):
            raise TypeError("swap01 expects bf16 or f32")
        x = x.contiguous()
        n0, n1, d = x.shape
        y = torch.empty((n1, n0, d), dtype=F32 if out_f32 else BF16, device=x.device)
        lib.check(lib.load().mmf_swap01(x.data_ptr(), y.data_ptr(), n0, n1, d, int(x.dtype == F32), int(out_f32),
                                        lib.stream_ptr()))
        ctx.in_f32 = x.dtype == F32
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n1, n0, d = g.shape
        dx = torch.empty((n0, n1, d), dtype=F32 if ctx.in_f32 else BF16, device=g.device)
        lib.check(lib.load().mmf_swap01(g.data_ptr(), dx.data_ptr(), n1, n0, d, int(g.dtype == F32), int(ctx.in_f32),
                                        lib.stream_ptr()))
        return dx, None


def swap01(x: torch.Tensor, out_f32: bool = False) -> torch.Tensor:
    return _Swap01.apply(x, out_f32)


def _workspace(dev) -> torch.Tensor:
    return torch.empty(lib.load().mmf_bilstm_workspace_bytes() // 4, dtype=torch.int32, device=dev)


class _BiLSTMLayer(torch.autograd.Function):
    """x_tb: bf16 (T*B, In) time-major.  params: w_ih, w_hh, b_ih, b_hh of direction 0, then of direction 1.
    Returns y_tb bf16 (T*B, 2H) — a view of the padded output buffer."""

    @staticmethod
    def forward(ctx, x_tb, T: int, B: int, *params):
        _req(x_tb, BF16)
        if B > MAX_B:
            raise ValueError(f"BiLSTM: batch {B} > {MAX_B} per launch (split the batch)")
        w_ih, w_hh, b_ih, b_hh = params[0::4], params[1::4], params[2::4], params[3::4]
        H = w_hh[0].shape[1]
        x_tb = x_tb.contiguous()
        if x_tb.shape[0] != T * B or w_ih[0].shape[1] != x_tb.shape[1]:
            raise ValueError(f"BiLSTM: x {tuple(x_tb.shape)} vs T*B={T * B}, input size {w_ih[0].shape[1]}")
        dev = x_tb.device
        gx = torch.empty((T * B, 8 * H), dtype=F32, device=dev)
        ops.gemm_group(GEMM_NT, [(x_tb, ops.shadow(w_ih[d]), gx[:, 4 * H * d:4 * H * (d + 1)], None, None) for d in range(2)], 0)
        ybuf = torch.zeros(((T + 2) * B, 2 * H), dtype=BF16, device=dev)        # zero row blocks in front and behind
        gates = torch.empty((T * B, 8 * H), dtype=F32, device=dev)
        cell = torch.empty((T * B, 2 * H), dtype=F32, device=dev)
        ws = _workspace(dev)
        P2 = C.c_void_p * 2
        args = lib.BiLstmArgs(gx.data_ptr(), P2(*[ops.shadow(w).data_ptr() for w in w_hh]),
                              P2(*[b.data_ptr() for b in b_ih]), P2(*[b.data_ptr() for b in b_hh]),
                              ybuf.data_ptr(), gates.data_ptr(), cell.data_ptr(), None, None, T, B, H)
        lib.check(lib.load().mmf_bilstm_layer_fwd(C.byref(args), ws.data_ptr(), ws.numel() * 4, lib.stream_ptr()))
        ctx.T, ctx.B, ctx.H = T, B, H
        ctx.params = params
        ctx.status = ws
        ctx.save_for_backward(x_tb, ybuf, gates, cell)
        ctx.x_needs = x_tb.requires_grad
        return ybuf[B:(T + 1) * B]

    @staticmethod
    def backward(ctx, dy):
        T, B, H = ctx.T, ctx.B, ctx.H
        x_tb, ybuf, gates, cell = ctx.saved_tensors
        params = ctx.params
        w_ih, w_hh, b_ih, b_hh = params[0::4], params[1::4], params[2::4], params[3::4]
        dev = dy.device
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = ops.cast_to_bf16(dy)
        dg = torch.empty((T * B, 8 * H), dtype=BF16, device=dev)
        ws = _workspace(dev)
        P2 = C.c_void_p * 2
        args = lib.BiLstmArgs(None, P2(*[ops.shadow(w).data_ptr() for w in w_hh]), P2(None, None), P2(None, None),
                              None, gates.data_ptr(), cell.data_ptr(), dy.data_ptr(), dg.data_ptr(), T, B, H)
        lib.check(lib.load().mmf_bilstm_layer_bwd(C.byref(args), ws.data_ptr(), ws.numel() * 4, lib.stream_ptr()))
        dgd = [dg[:, 4 * H * d:4 * H * (d + 1)] for d in range(2)]
        dx = None
        if ctx.x_needs:                                  # dx = dG_0 W_ih_0 + dG_1 W_ih_1
            dx = torch.empty(x_tb.shape, dtype=BF16, device=dev)
            ops.gemm(GEMM_NN, dgd[0], ops.shadow(w_ih[0]), dx)
            ops.gemm(GEMM_NN, dgd[1], ops.shadow(w_ih[1]), dx, aux=dx, epilogue=EPI_ADD_AUX)
        # h_{t-1} as the steps saw it: direction 0 reads the row block before, direction 1 the one after (zeros at the ends)
        hprev = [ybuf[0:T * B, 0:H], ybuf[2 * B:(T + 2) * B, H:2 * H]]
        for d in range(2):
            ops.queue_wgrad(dgd[d], x_tb, W(w_ih[d]).grad, W(b_ih[d]).grad)
            ops.queue_wgrad(dgd[d], hprev[d], W(w_hh[d]).grad, W(b_hh[d]).grad)
        return (dx, None, None) + (None,) * len(params)


def bilstm(lstm: torch.nn.LSTM, x: torch.Tensor, dropout_p: float = 0.0) -> torch.Tensor:
    """``lstm(x)[0]`` for a batch-first bidirectional ``nn.LSTM`` (zero initial state): x (B, T, In) f32 or bf16 ->
    (B, T, 2H) bf16.  ``dropout_p``: the inter-layer dropout of ``nn.LSTM(dropout=...)`` (training only, applied to
    the outputs of every layer but the last)."""
    if not (lstm.bidirectional and lstm.batch_first and lstm.bias and lstm.proj_size == 0):
        raise ValueError("bilstm: expects nn.LSTM(batch_first=True, bidirectional=True, bias=True)")
    if not x.is_cuda:
        raise RuntimeError("mmfusion LSTM runs on the GPU only (no CPU fallback)")
    if ops.fp32_mode():
        # the persistent recurrence keeps W_hh as bf16 MFMA fragments; there is no f32 form of it (the bf16 GEMM
        # wrappers used to fail on the f32 masters with an unrelated TypeError)
        raise RuntimeError("the fp32 parity mode does not cover the BiLSTM (video encoder): run it with precision='bf16'")
    B, T, In = x.shape
    outs: List[torch.Tensor] = []
    for b0 in range(0, B, MAX_B):                                   # samples are independent: chunks of <= 64
        xb = x[b0:b0 + MAX_B]
        Bc = xb.shape[0]
        h = swap01(xb).view(T * Bc, In)                             # time-major bf16 rows
        for layer in range(lstm.num_layers):
            params: List[torch.nn.Parameter] = []
            for suffix in ("", "_reverse"):
                params += [getattr(lstm, f"{n}_l{layer}{suffix}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            h = _BiLSTMLayer.apply(h, T, Bc, *params)
            if layer + 1 < lstm.num_layers:
                h = ops.dropout(h, dropout_p, dropout_p > 0.0)
        outs.append(swap01(h.reshape(T, Bc, -1)))                   # back to (B, T, 2H)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


def last_status(ctx_ws: torch.Tensor) -> int:
    """status word of a BiLSTM launch's workspace (0 = ok, 1 = a workgroup gave up waiting; synchronises)."""
    return int(ctx_ws[2].item())
