import os, sys
sys.path.insert(0, "simple-multimodal_amd"); sys.path.insert(0, "tools")
import torch
from mmfusion import lib
from mmfusion.lib import GEMM_NT
from gemm_bench import bench
L = lib.load()
for impl in (4, 6):
    lib.check(L.mmf_gemm_select_impl(impl))
    for (M, N, K) in [(1024, 1024, 64), (2048, 2048, 64), (4096, 4096, 64), (8192, 3072, 64), (8192, 3072, 128), (8192, 3072, 768), (2048, 2048, 768), (1024, 1024, 768)]:
        us, tf = bench(GEMM_NT, [(M, N, K)], reps=20)
        mb = M * N * 2 / 1e6
        print(f"impl{impl} {M}x{N}x{K}: {us:7.1f} us  C = {mb:.0f} MB -> {mb / us * 1e3 / 1e3:5.2f} TB/s of output", flush=True)
lib.check(L.mmf_gemm_select_impl(0))
