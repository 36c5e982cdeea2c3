#!/usr/bin/env python3
"""Where a wave of the backward attention kernels spends its cycles (build: tools/build_variant.sh stamps attention2.hip
-DMMF_ATTN_STAMPS; run with MMF_LIB_PATH on that library; measurement only).  Segments per wave (s_memtime cycles, 100 MHz
... no: shader clock): prologue, vmcnt wait, barrier, DMA issue, S/dP products, P/dS arithmetic, chain (dQ^T or dV^T/dK^T), epilogue."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib

L = lib.load()
fn = C.CDLL(L._name).mmf_debug_attn2_stamps
fn.argtypes = [C.c_void_p]
B, H, dh, d = int(os.environ.get("MMF_ATTN_B", 16)), 8, 96, 768
Tq, Tk = int(os.environ.get("TQ", 512)), int(os.environ.get("TK", 400))
Q = torch.randn(B * Tq, d, device="cuda").bfloat16()
KV = torch.randn(B * Tk, 2 * d, device="cuda").bfloat16()
O, dO, dQ, dKV = torch.empty_like(Q), torch.randn_like(Q), torch.empty_like(Q), torch.empty_like(KV)
LSE, delta = torch.empty(B * H * Tq, device="cuda"), torch.empty(B * H * Tq, device="cuda")
arr = (lib.AttnProblem * 1)()
p = arr[0]
p.Q, p.K, p.V, p.O, p.LSE = Q.data_ptr(), KV.data_ptr(), KV.data_ptr() + 2 * d, O.data_ptr(), LSE.data_ptr()
p.dO, p.delta, p.dQ, p.dK, p.dV = dO.data_ptr(), delta.data_ptr(), dQ.data_ptr(), dKV.data_ptr(), dKV.data_ptr() + 2 * d
p.B, p.H, p.Tq, p.Tk = B, H, Tq, Tk
p.ldq, p.ldk, p.ldv, p.ldo = d, 2 * d, 2 * d, d
sc = dh ** -0.5
lib.check(L.mmf_attn_fwd_grouped(arr, 1, dh, sc, lib.stream_ptr()))
for _ in range(3):
    lib.check(L.mmf_attn_bwd_grouped(arr, 1, dh, sc, lib.stream_ptr()))
torch.cuda.synchronize()
buf = torch.zeros(2 * 65536 * 4 * 12, dtype=torch.int64, device="cuda")
assert fn(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.check(L.mmf_attn_bwd_grouped(arr, 1, dh, sc, lib.stream_ptr()))
e1.record()
torch.cuda.synchronize()
fn(None)
names = ["prologue", "vmcnt", "barrier", "dma", "S/dP", "arith", "chain", "epilogue"]
print(f"Tq {Tq} Tk {Tk} B {B}: dQ + dK/dV launches {e0.elapsed_time(e1) * 1e3:.1f} us (event-timed, stamped build)")
s = buf.view(2, 65536, 4, 12).double()
for k, (kname, sweep, part) in enumerate((("dQ", Tk, Tq), ("dK/dV", Tq, Tk))):
    live = s[k][:, :, 0] > 0                       # waves that stored stamps
    v = s[k][live]
    if v.numel() == 0:
        continue
    m = v.mean(dim=0)
    tot = m[:8].sum().item()
    nblk = ((sweep + 31) // 32)
    print(f"  {kname:6s} {int(live.sum())} active waves, mean cycles per wave {tot:8.0f}: " +
          "  ".join(f"{names[i]} {m[i].item():7.0f}" for i in range(8)))
    print(f"         per 32-row block of the sweep ({nblk} blocks): S/dP {m[4].item() / nblk:6.0f}  arith {m[5].item() / nblk:6.0f}  "
          f"chain {m[6].item() / nblk:6.0f};  per 64-row tile: vmcnt {m[1].item() / ((sweep + 63) // 64):6.0f}  "
          f"barrier {m[2].item() / ((sweep + 63) // 64):6.0f}  dma {m[3].item() / ((sweep + 63) // 64):6.0f}")
    mx = v[:, :8].sum(dim=1)
    print(f"         wave total: min {mx.min().item():.0f}  median {mx.median().item():.0f}  max {mx.max().item():.0f}")
