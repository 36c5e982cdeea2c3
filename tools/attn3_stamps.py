#!/usr/bin/env python3
"""Where a wave of the one-wave-per-SIMD attention forward (attention3.hip) spends its cycles.  Build: tools/build_variant.sh stamps3
attention3.hip -DMMF_ATTN_STAMPS; run with MMF_LIB_PATH on that library and MMF_ATTN_FWD_GEN=3; measurement only (every stamp is an
s_memtime and drains the wave's outstanding LDS reads)."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib

L = lib.load()
fn = C.CDLL(L._name).mmf_debug_attn3_stamps
fn.argtypes = [C.c_void_p]
B, H, dh, d = int(os.environ.get("MMF_ATTN_B", 16)), 8, 96, 768
Tq, Tk = int(os.environ.get("TQ", 512)), int(os.environ.get("TK", 512))
Q = torch.randn(B * Tq, d, device="cuda").bfloat16()
KV = torch.randn(B * Tk, 2 * d, device="cuda").bfloat16()
O = torch.empty_like(Q)
LSE = torch.empty(B * H * Tq, device="cuda")
arr = (lib.AttnProblem * 1)()
p = arr[0]
p.Q, p.K, p.V, p.O, p.LSE = Q.data_ptr(), KV.data_ptr(), KV.data_ptr() + 2 * d, O.data_ptr(), LSE.data_ptr()
p.B, p.H, p.Tq, p.Tk = B, H, Tq, Tk
p.ldq, p.ldk, p.ldv, p.ldo = d, 2 * d, 2 * d, d
sc = dh ** -0.5
for _ in range(3):
    lib.check(L.mmf_attn_fwd_grouped(arr, 1, dh, sc, lib.stream_ptr()))
torch.cuda.synchronize()
buf = torch.zeros(65536 * 4 * 8, dtype=torch.int64, device="cuda")
assert fn(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.check(L.mmf_attn_fwd_grouped(arr, 1, dh, sc, lib.stream_ptr()))
e1.record()
torch.cuda.synchronize()
fn(None)
names = ["prologue", "S(0)", "rescale", "phase1", "middle", "phase2", "epilogue"]
s = buf.view(65536, 4, 8).double()
live = s[:, :, 7] > 0
v = s[live]
m = v.mean(dim=0)
nblk = int(m[7].item())
print(f"Tq {Tq} Tk {Tk} B {B}: forward launch {e0.elapsed_time(e1) * 1e3:.1f} us (event-timed, stamped build); {int(live.sum())} active waves, {nblk} blocks")
print("  mean cycles per wave " + f"{m[:7].sum().item():8.0f}: " + "  ".join(f"{names[i]} {m[i].item():7.0f}" for i in range(7)))
print(f"  per 32-key block: rescale {m[2].item() / nblk:6.0f}  phase1 {m[3].item() / nblk:6.0f}  middle {2 * m[4].item() / nblk:6.0f} (per tile)  phase2 {m[5].item() / nblk:6.0f}")
tot = v[:, :7].sum(dim=1)
print(f"  wave total: min {tot.min().item():.0f}  median {tot.median().item():.0f}  max {tot.max().item():.0f}")
