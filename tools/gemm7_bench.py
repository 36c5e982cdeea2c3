#!/usr/bin/env python3
"""generation 6 (one tile per workgroup) vs generation 7 (persistent workgroups) on the MulT step's NT / NN launch groups,
interleaved in one process (HIP events, graph-free back-to-back launches):  python tools/gemm7_bench.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch  # noqa: E402
from mmfusion import lib, ops  # noqa: E402
from mmfusion.lib import EPI_ADD_AUX, EPI_BIAS, EPI_MASK_AUX, EPI_RELU, GEMM_NN, GEMM_NT  # noqa: E402

DEV = "cuda"


def make(layout, shapes, epi):
    probs = []
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=DEV).bfloat16()
        B = (torch.randn(N, K, device=DEV) if layout == GEMM_NT else torch.randn(K, N, device=DEV)).bfloat16() * K ** -0.5
        C = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        bias = torch.randn(N, device=DEV) if epi & EPI_BIAS else None
        aux = torch.randn(M, N, device=DEV).bfloat16() if epi & (EPI_ADD_AUX | EPI_MASK_AUX) else None
        probs.append((A, B, C, bias, aux))
    return probs


def time_impl(L, impl, layout, probs, epi, reps):
    lib.check(L.mmf_gemm_select_impl(impl))
    for _ in range(2):
        ops.gemm_group(layout, probs, epi)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_group(layout, probs, epi)
    e1.record()
    torch.cuda.synchronize()
    lib.check(L.mmf_gemm_select_impl(0))
    return e0.elapsed_time(e1) * 1e3 / reps


def time_cold(L, impl, layout, probs, epi, reps, flush):
    """one launch at a time between events; flush: a 1 GiB fill in front of every launch (L2 and Infinity Cache hold nothing of the operands)"""
    lib.check(L.mmf_gemm_select_impl(impl))
    junk = torch.empty(1 << 28, device=DEV, dtype=torch.float32) if flush else None
    tot = 0.0
    for _ in range(reps):
        if flush:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ops.gemm_group(layout, probs, epi)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    lib.check(L.mmf_gemm_select_impl(0))
    return tot / reps


def main():
    L = lib.load()
    big, aud = 8192, 6400
    cases = [
        ("NT 4096^3", GEMM_NT, [(4096, 4096, 4096)], 0),
        ("NT 8192^2 x 768", GEMM_NT, [(8192, 8192, 768)], 0),
        ("NT in-proj text (l7)", GEMM_NT, [(big, 768, 768)] * 2 + [(aud, 1536, 768), (480, 1536, 768)], EPI_BIAS),
        ("NT in-proj a+v (l0)", GEMM_NT, [(aud, 768, 768)] * 2 + [(big, 1536, 768)] * 2 + [(480, 1536, 768)] + [(480, 768, 768)] * 2 + [(aud, 1536, 768)], EPI_BIAS),
        ("NT out-proj text (l9)", GEMM_NT, [(big, 768, 768)] * 2, EPI_BIAS | EPI_ADD_AUX),
        ("NT out-proj a+v (l2)", GEMM_NT, [(aud, 768, 768)] * 2 + [(480, 768, 768)] * 2, EPI_BIAS | EPI_ADD_AUX),
        ("NT ffn1 text (l10)", GEMM_NT, [(big, 3072, 768)] * 2, EPI_BIAS | EPI_RELU),
        ("NT ffn1 a+v (l3)", GEMM_NT, [(aud, 3072, 768)] * 2 + [(480, 3072, 768)] * 2, EPI_BIAS | EPI_RELU),
        ("NT ffn2 text (l11)", GEMM_NT, [(big, 768, 3072)] * 2, EPI_BIAS | EPI_ADD_AUX),
        ("NT ffn2 a+v (l4)", GEMM_NT, [(aud, 768, 3072)] * 2 + [(480, 768, 3072)] * 2, EPI_BIAS | EPI_ADD_AUX),
        ("NT self qkv text (l12)", GEMM_NT, [(big, 2304, 768)], EPI_BIAS),
        ("NT self qkv a+v (l5)", GEMM_NT, [(aud, 2304, 768), (480, 2304, 768)], EPI_BIAS),
        ("NN dH text (l16)", GEMM_NN, [(big, 3072, 768)] * 2, EPI_MASK_AUX),
        ("NN dX text (l17)", GEMM_NN, [(big, 768, 3072)] * 2, EPI_ADD_AUX),
        ("NN d-out-proj text (l18)", GEMM_NN, [(big, 768, 768)] * 2, 0),
        ("NN self qkv dgrad (l15)", GEMM_NN, [(big, 768, 2304)], 0),
        ("NN self qkv dgrad a+v (l22)", GEMM_NN, [(aud, 768, 2304), (480, 768, 2304)], 0),
        ("NN in-proj dgrad text (l20)", GEMM_NN, [(big, 768, 768)] * 2 + [(aud, 768, 1536), (480, 768, 1536)], 0),
        ("NN in-proj dgrad a+v (l27)", GEMM_NN, [(aud, 768, 768)] * 2 + [(big, 768, 1536)] * 2 + [(480, 768, 1536)] + [(480, 768, 768)] * 2 + [(aud, 768, 1536)], 0),
    ]
    impls = [int(x) for x in os.environ.get("IMPLS", "6,7").split(",")]
    reps = int(os.environ.get("REPS", "20"))
    print(f"{'case':30s} tiles " + " ".join(f"{'g%d us' % i:>9s} {'TF':>7s}" for i in impls))
    only = os.environ.get("ONLY")
    for name, layout, shapes, epi in cases:
        if only and only not in name:
            continue
        probs = make(layout, shapes, epi)
        fl = sum(2.0 * M * N * K for M, N, K in shapes)
        tiles = sum(((M + 255) // 256) * ((N + 255) // 256) for M, N, K in shapes)
        best = {i: float("inf") for i in impls}
        for _ in range(3):                                         # interleaved rounds, best of three
            for i in impls:
                best[i] = min(best[i], time_impl(L, i, layout, probs, epi, reps))
        row = f"{name:30s} {tiles:5d} " + " ".join(f"{best[i]:9.1f} {fl / best[i] / 1e6:7.0f}" for i in impls)
        if os.environ.get("COLD"):                                 # single launches: operands warm / flushed out of the caches
            row += "   single warm " + " ".join(f"{time_cold(L, i, layout, probs, epi, 8, False):7.1f}" for i in impls)
            row += "   single cold " + " ".join(f"{time_cold(L, i, layout, probs, epi, 8, True):7.1f}" for i in impls)
        print(row, flush=True)
        del probs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
