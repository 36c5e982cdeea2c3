#!/usr/bin/env python3
"""Main-loop ablations of the v2 GEMM (debug epilogue bits 1<<29: no DMA after the prologue,
1<<28: no LDS fragment reads after the first k-step).  Results are wrong by construction; only time matters."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd")); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
from mmfusion.lib import GEMM_NT, GEMM_NN, GEMM_TN, EPI_ACCUM
from gemm_bench import bench
rows = [8192, 8192, 6400, 6400, 480, 480]
for name, layout, shapes, f32, base in [("NT 4096^3", GEMM_NT, [(4096, 4096, 4096)], False, 0),
                                 ("NT ffn1 x6", GEMM_NT, [(r, 3072, 768) for r in rows], False, 0),
                                 ("NT ffn2 x6", GEMM_NT, [(r, 768, 3072) for r in rows], False, 0),
                                 ("TN 4096^3", GEMM_TN, [(4096, 4096, 4096)], True, EPI_ACCUM)]:
    for tag, bits in [("full", 0), ("noDMA", 1 << 29), ("noLDS", 1 << 28), ("noDMA+noLDS", (1 << 29) | (1 << 28))]:
        us, tf = bench(layout, shapes, epi=base | bits, f32=f32)
        print(f"{name:12s} {tag:12s} {us:9.1f} us {tf:8.1f} TF(eq)", flush=True)
