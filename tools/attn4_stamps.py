#!/usr/bin/env python3
"""Where a wave of the fourth-generation attention forward spends its cycles (build: make EXTRA=-DMMF_ATTN_STAMPS; measurement
only).  Segments: 0 prologue, 1 matrix phases, 2 softmax phases, 3 vmcnt waits, 4 barriers, 5 DMA issue, 6 epilogue."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib

L = lib.load()
fn = C.CDLL(L._name).mmf_debug_attn4_stamps
fn.argtypes = [C.c_void_p]
B, H, dh, d = int(os.environ.get("MMF_ATTN_B", 16)), 8, 96, 768
Tq, Tk = int(os.environ.get("TQ", 512)), int(os.environ.get("TK", 400))
Q = torch.randn(B * Tq, d, device="cuda").bfloat16()
KV = torch.randn(B * Tk, 2 * d, device="cuda").bfloat16()
O = torch.empty_like(Q)
LSE = torch.empty(B * H * Tq, device="cuda")
arr = (lib.AttnProblem * 1)()
p = arr[0]
p.Q, p.K, p.V, p.O, p.LSE = Q.data_ptr(), KV.data_ptr(), KV.data_ptr() + 2 * d, O.data_ptr(), LSE.data_ptr()
p.B, p.H, p.Tq, p.Tk = B, H, Tq, Tk
p.ldq, p.ldk, p.ldv, p.ldo = d, 2 * d, 2 * d, d
lib.check(L.mmf_attn_select_impl(4))
nwg = B * H * ((Tq + 255) // 256)
buf = torch.zeros((nwg + 8) * 8 * 12, dtype=torch.int64, device="cuda")
for _ in range(3):
    lib.check(L.mmf_attn_fwd_grouped(arr, 1, dh, dh ** -0.5, lib.stream_ptr()))
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.check(L.mmf_attn_fwd_grouped(arr, 1, dh, dh ** -0.5, lib.stream_ptr()))
e1.record()
torch.cuda.synchronize()
fn(None)
s = buf.view(-1, 8, 12)[:nwg].double()
n = (Tk + 63) // 64
names = ["prologue", "matrix", "softmax", "vmcnt", "barrier", "dma", "epilogue", "-"]
print(f"Tq {Tq} Tk {Tk} B {B}: {e0.elapsed_time(e1) * 1e3:.1f} us (one launch, event-timed); {n} tiles; cycles per wave (mean over workgroups)")
for g, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    v = s[:, sl].mean(dim=(0, 1))
    tot = v[:7].sum().item() + v[8:].sum().item()
    print(f"  {g}: total {tot:8.0f} | " + "  ".join(f"{names[i]} {v[i].item():7.0f}" for i in range(7)))
    print(f"           softmax phase split: max+decide {v[8].item() / n:6.0f}  probs(block 0) {v[9].item() / n:6.0f}  probs(block 1) {v[10].item() / n:6.0f}  (rest {v[2].item() / n:5.0f});  matrix phase split: P.V {v[11].item() / n:6.0f}  QK^T+rest {v[1].item() / n:6.0f}")
    print(f"           per phase: matrix {(v[1].item() + v[11].item()) / (n + 1):6.0f}  softmax {(v[2].item() + v[8].item() + v[9].item() + v[10].item()) / n:6.0f}  barrier {v[4].item() / (2 * n):6.0f}  vmcnt {v[3].item() / n:6.0f}")
lib.check(L.mmf_attn_select_impl(0))
hw = buf.view(-1, 8, 12)[:nwg, :, 7]
for wg in (0, 1, 2, 17):
    print(f"  workgroup {wg}: SIMD of waves 0..7 = {[(int(x) >> 4) & 3 for x in hw[wg].tolist()]}  CU = {[(int(x) >> 8) & 15 for x in hw[wg].tolist()][:2]}")
