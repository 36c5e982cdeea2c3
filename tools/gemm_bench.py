#!/usr/bin/env python3
"""GEMM microbenchmark (GPU box): TFLOP/s of mmf_gemm_grouped per layout / shape, HIP-event timed.
    MMF_GEMM_IMPL=1|2 python tools/gemm_bench.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch  # noqa: E402
from mmfusion import ops  # noqa: E402
from mmfusion.lib import GEMM_NN, GEMM_NT, GEMM_TN, EPI_ACCUM, EPI_BIAS  # noqa: E402

DEV = "cuda"


def bench(layout, shapes, reps=20, epi=0, f32=False):
    probs = []
    for (M, N, K) in shapes:
        if layout == GEMM_NT:
            A, B = torch.randn(M, K, device=DEV).bfloat16(), torch.randn(N, K, device=DEV).bfloat16()
        elif layout == GEMM_NN:
            A, B = torch.randn(M, K, device=DEV).bfloat16(), torch.randn(K, N, device=DEV).bfloat16()
        else:
            A, B = torch.randn(K, M, device=DEV).bfloat16(), torch.randn(K, N, device=DEV).bfloat16()
        C = torch.zeros(M, N, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
        bias = torch.randn(N, device=DEV) if epi & EPI_BIAS else None
        probs.append((A, B, C, bias, None))
    for _ in range(3):
        ops.gemm_group(layout, probs, epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_group(layout, probs, epi)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = sum(2.0 * M * N * K for M, N, K in shapes)
    return us, fl / us / 1e6


def main():
    from mmfusion import lib
    L = lib.load()
    impls = [int(x) for x in os.environ.get("IMPLS", "2,6,7").split(",")]
    rows = [8192, 8192, 6400, 6400, 480, 480]
    cases = [
        ("NT 4096^3", GEMM_NT, [(4096, 4096, 4096)], 0, False),
        ("NT 8192x3072x768", GEMM_NT, [(8192, 3072, 768)], 0, False),
        ("NT 8192x768x768", GEMM_NT, [(8192, 768, 768)], 0, False),
        ("NT ffn1 group x6", GEMM_NT, [(r, 3072, 768) for r in rows], EPI_BIAS, False),
        ("NT ffn2 group x6", GEMM_NT, [(r, 768, 3072) for r in rows], EPI_BIAS, False),
        ("NT outproj group x6", GEMM_NT, [(r, 768, 768) for r in rows], EPI_BIAS, False),
        ("NT inproj group x12", GEMM_NT, [(r, 768, 768) for r in rows] + [(r, 1536, 768) for r in rows], EPI_BIAS, False),
        ("NT self inproj x3", GEMM_NT, [(r, 2304, 768) for r in (8192, 6400, 480)], EPI_BIAS, False),
        ("NT ffn1 group x3", GEMM_NT, [(r, 3072, 768) for r in (8192, 6400, 480)], EPI_BIAS, False),
        ("NT ffn2 group x3", GEMM_NT, [(r, 768, 3072) for r in (8192, 6400, 480)], EPI_BIAS, False),
        ("NT outproj group x3", GEMM_NT, [(r, 768, 768) for r in (8192, 6400, 480)], EPI_BIAS, False),
        ("NT inproj group x6", GEMM_NT, [(8192, 768, 768), (6400, 1536, 768), (6400, 768, 768), (480, 1536, 768), (480, 768, 768), (8192, 1536, 768)], EPI_BIAS, False),
        ("NN 4096^3", GEMM_NN, [(4096, 4096, 4096)], 0, False),
        ("NN dH group x6", GEMM_NN, [(r, 3072, 768) for r in rows], 0, False),
        ("NN dX group x6", GEMM_NN, [(r, 768, 3072) for r in rows], 0, False),
        ("NN outproj dgrad x6", GEMM_NN, [(r, 768, 768) for r in rows], 0, False),
        ("NN inproj dgrad x12", GEMM_NN, [(r, 768, 768) for r in rows] + [(r, 768, 1536) for r in rows], 0, False),
        ("NN self inproj dgrad x3", GEMM_NN, [(r, 768, 2304) for r in (8192, 6400, 480)], 0, False),
        ("NN dH group x3", GEMM_NN, [(r, 3072, 768) for r in (8192, 6400, 480)], 0, False),
        ("NN dX group x3", GEMM_NN, [(r, 768, 3072) for r in (8192, 6400, 480)], 0, False),
        ("NN outproj dgrad x3", GEMM_NN, [(r, 768, 768) for r in (8192, 6400, 480)], 0, False),
        ("NN inproj dgrad x6", GEMM_NN, [(8192, 768, 768), (6400, 768, 1536), (6400, 768, 768), (480, 768, 1536), (480, 768, 768), (8192, 768, 1536)], 0, False),
        ("TN 4096^3", GEMM_TN, [(4096, 4096, 4096)], EPI_ACCUM, True),
        ("TN dW1 group x6", GEMM_TN, [(3072, 768, r) for r in rows], EPI_ACCUM, True),
        ("TN dWo group x6", GEMM_TN, [(768, 768, r) for r in rows], EPI_ACCUM, True),
        # every weight gradient of the six cross blocks, as the deferred launch holds them (q rows, kv rows per block)
        ("TN cross blocks x30", GEMM_TN, [s for q, kv in ((8192, 6400), (8192, 480), (6400, 8192), (6400, 480), (480, 8192), (480, 6400))
                                           for s in ((768, 768, q), (1536, 768, kv), (768, 768, q), (3072, 768, q), (768, 3072, q))], EPI_ACCUM, True),
    ]
    if os.environ.get("CASES"):
        cases = [c for c in cases if c[0].split()[0] in os.environ["CASES"].split(",")]
    rounds = int(os.environ.get("ROUNDS", "3"))
    for name, layout, shapes, epi, f32 in cases:
        best = {i: 0.0 for i in impls}
        for _ in range(rounds):                       # interleaved rounds in ONE process (guide rule 24)
            for i in impls:
                lib.check(L.mmf_gemm_select_impl(i))
                us, tf = bench(layout, shapes, reps=10, epi=epi, f32=f32)
                best[i] = max(best[i], tf)
        print(f"{name:24s} " + "  ".join(f"impl{i} {best[i]:7.1f} TF" for i in impls), flush=True)
    lib.check(L.mmf_gemm_select_impl(0))


if __name__ == "__main__":
    main()
