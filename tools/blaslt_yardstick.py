#!/usr/bin/env python3
"""Vendor-library yardstick (not used by the product path): torch.matmul (hipBLASLt) bf16 on the MulT GEMM shapes,
next to mmf_gemm_grouped on the same single problem.  Prints TFLOP/s for both."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib, ops

def timeit(fn, reps=20, inner=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)

shapes = [("out-proj 8192x768x768", 8192, 768, 768), ("in-proj kv 8192x1536x768", 8192, 1536, 768),
          ("ffn1 8192x3072x768", 8192, 3072, 768), ("ffn2 8192x768x3072", 8192, 768, 3072),
          ("self qkv 8192x2304x768", 8192, 2304, 768), ("4096^3", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wt = w.t()
    t_lt = timeit(lambda: torch.matmul(x, wt, out=y))
    t_nt = timeit(lambda: ops.gemm_group(lib.GEMM_NT, [(x, w, y, None, None)], 0))
    # wgrad shape: dW[N][K] = dy[M][N]^T x[M][K]
    dy = torch.randn(M, N, device="cuda").bfloat16()
    dw = torch.empty(N, K, device="cuda", dtype=torch.float32)
    dwb = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    t_lt_tn = timeit(lambda: torch.matmul(dy.t(), x, out=dwb))
    t_tn = timeit(lambda: ops.gemm_group(lib.GEMM_TN, [(dy, x, dw, None, None)], 0))
    fl = 2.0 * M * N * K / 1e6
    print(f"{name:28s} NT: hipBLASLt {fl / t_lt:7.1f} TF  mmf {fl / t_nt:7.1f} TF   |  TN (wgrad): hipBLASLt {fl / t_lt_tn:7.1f} TF  mmf {fl / t_tn:7.1f} TF", flush=True)
