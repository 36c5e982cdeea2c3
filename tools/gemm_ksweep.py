#!/usr/bin/env python3
"""time vs K at fixed M x N: slope = per-k-step cost, intercept = per-tile overhead."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd")); sys.path.insert(0, os.path.join(REPO, "tools"))
from mmfusion import lib
from mmfusion.lib import GEMM_NT, GEMM_NN
from gemm_bench import bench
L = lib.load()
for (M, N) in [(8192, 3072), (8192, 768), (8192, 2048)]:
    for impl in (2, 4):
        lib.check(L.mmf_gemm_select_impl(impl))
        row = []
        for K in (64, 128, 256, 512, 768, 1536, 3072, 6144):
            best = min(bench(GEMM_NT, [(M, N, K)], reps=10)[0] for _ in range(3))
            row.append((K, best))
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        print(f"M={M} N={N} tiles={tiles} impl{impl}: " + "  ".join(f"K{k}:{t:7.1f}us" for k, t in row), flush=True)
lib.check(L.mmf_gemm_select_impl(2))
