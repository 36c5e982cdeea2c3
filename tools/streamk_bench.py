#!/usr/bin/env python3
"""Stream-K vs data-parallel on the MulT launch groups (256 x 256 kernel pinned, same process, hipGraph of 10 launches).
    python tools/streamk_bench.py"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib, ops
from mmfusion.lib import EPI_ADD_AUX, EPI_BIAS, EPI_MASK_AUX, EPI_RELU, GEMM_NN, GEMM_NT

GROUPS = [
    ("out-proj NT  K=768  N=768 ", GEMM_NT, [(8192, 768, 768), (6400, 768, 768), (480, 768, 768)], EPI_BIAS | EPI_ADD_AUX),
    ("FFN1 NT      K=768  N=3072", GEMM_NT, [(8192, 3072, 768), (6400, 3072, 768), (480, 3072, 768)], EPI_BIAS | EPI_RELU),
    ("FFN2 NT      K=3072 N=768 ", GEMM_NT, [(8192, 768, 3072), (6400, 768, 3072), (480, 768, 3072)], EPI_BIAS | EPI_ADD_AUX),
    ("in-proj NT   K=768        ", GEMM_NT, [(8192, 768, 768), (480, 1536, 768), (6400, 768, 768), (8192, 1536, 768), (480, 768, 768), (6400, 1536, 768)], EPI_BIAS),
    ("self in-proj K=768  N=2304", GEMM_NT, [(8192, 2304, 768), (6400, 2304, 768), (480, 2304, 768)], EPI_BIAS),
    ("dH NN        K=768  N=3072", GEMM_NN, [(8192, 3072, 768), (6400, 3072, 768), (480, 3072, 768)], EPI_MASK_AUX),
    ("dX NN        K=3072 N=768 ", GEMM_NN, [(8192, 768, 3072), (6400, 768, 3072), (480, 768, 3072)], EPI_ADD_AUX),
    ("dqkv NN      K=2304 N=768 ", GEMM_NN, [(8192, 768, 2304), (6400, 768, 2304), (480, 768, 2304)], 0),
    ("text pair NT K=3072 N=768 ", GEMM_NT, [(8192, 768, 3072), (8192, 768, 3072)], EPI_BIAS | EPI_ADD_AUX),
]


def build(layout, shapes, epi):
    probs = []
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        Bm = (torch.randn(N, K, device="cuda") if layout == GEMM_NT else torch.randn(K, N, device="cuda")).bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda") if epi & EPI_BIAS else None
        aux = torch.randn(M, N, device="cuda").bfloat16() if epi & (EPI_ADD_AUX | EPI_MASK_AUX) else None
        probs.append((A, Bm, C, bias, aux))
    return probs


def timed(fn, inner=10, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (inner * reps)


ops.streamk_workspace(torch.device("cuda", 0))          # allocate the eager slot before any capture
for name, layout, shapes, epi in GROUPS:
    probs = build(layout, shapes, epi)
    fl = sum(2.0 * M * N * K for M, N, K in shapes)
    row = f"{name}"
    for label, impl, sk in (("auto/DP", 0, "0"), ("256x256 DP", 4, "0"), ("256x256 stream-K", 4, "1")):
        os.environ["MMF_GEMM_STREAMK"] = sk
        lib.check(lib.load().mmf_gemm_select_impl(impl))
        us = timed(lambda: ops.gemm_group(layout, probs, epi))
        row += f"   {label} {us:6.1f} us {fl / us / 1e6:6.0f} TF"
    lib.check(lib.load().mmf_gemm_select_impl(0))
    print(row, flush=True)
