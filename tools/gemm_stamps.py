#!/usr/bin/env python3
"""Where a wave of the 256x256 GEMM spends its cycles (build with `make EXTRA=-DMMF_GEMM_STAMPS`; measurement only).
Segments per k-step: 0 vmcnt wait, 1 barrier, 2 DMA issue (waves 0-3), 3 compute k-half 0, 4 DMA issue (waves 4-7),
5 compute k-half 1; 6 prologue, 7 epilogue.  Prints cycles per k-step (0-5) or per tile (6, 7), averaged per wave group."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib, ops
from mmfusion.lib import GEMM_NT, GEMM_NN, GEMM_TN

L = lib.load()
L.mmf_gemm_select_impl(4)
fn = C.CDLL(L._name).mmf_debug_gemm4_stamps
fn.argtypes = [C.c_void_p, C.c_int]
for name, layout, (M, N, K) in [("NT 4096^3", GEMM_NT, (4096, 4096, 4096)), ("NT 8192x3072x768", GEMM_NT, (8192, 3072, 768)),
                                ("NN 8192x768x3072", GEMM_NN, (8192, 768, 3072)), ("NN 4096^3", GEMM_NN, (4096, 4096, 4096))]:
    if layout == GEMM_NT:
        A, B = torch.randn(M, K, device="cuda").bfloat16(), torch.randn(N, K, device="cuda").bfloat16()
    else:
        A, B = torch.randn(M, K, device="cuda").bfloat16(), torch.randn(K, N, device="cuda").bfloat16()
    Cm = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(ntile * 8 * 8, dtype=torch.int64, device="cuda")
    # MMF_STAMPS_NOMEM=1: leading dimensions 0 -> empty buffer ranges, every LDS-DMA piece returns zeros without touching
    # L2 / HBM: the k-loop's issue / LDS / MFMA-bound rate
    nomem = bool(os.environ.get("MMF_STAMPS_NOMEM"))
    prob = [lib.GemmProblem(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, None, M, N, K, A.stride(0), B.stride(0), N, 0)]

    def go():
        lib.gemm_grouped(prob, layout, 0, False, 1.0)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    assert fn(buf.data_ptr(), int(nomem)) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    fn(None, 0)
    us = e0.elapsed_time(e1) * 1e3
    s = buf.view(ntile, 8, 8).double()
    nk = (K + 63) // 64
    lo, hi = s[:, :4].mean(dim=(0, 1)), s[:, 4:].mean(dim=(0, 1))
    tot = s.sum(dim=2).mean().item()
    print(f"{name:18s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF  wave lifetime {tot:9.0f} cyc = {tot / nk:7.0f} per k-step (MFMA floor 2048 per SIMD)")
    for tag, v in (("waves 0-3", lo), ("waves 4-7", hi)):
        print(f"   {tag}: per k-step  wait {v[0] / nk:6.0f}  barrier {v[1] / nk:6.0f}  dma {v[2] / nk:6.0f}  mma0 {v[3] / nk:6.0f}  dma {v[4] / nk:6.0f}  mma1 {v[5] / nk:6.0f}"
              f"   | prologue {v[6]:7.0f}  epilogue {v[7]:7.0f}")
