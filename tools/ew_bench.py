#!/usr/bin/env python3
"""Streaming-kernel microbench: f32->bf16 cast of the MulT weight arena, LayerNorm fwd/bwd, add3, meanpool at the
MulT shapes; prints achieved HBM GB/s of algorithmic bytes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib, ops

def timeit(fn, reps=20, inner=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)

L = lib.load()
n = 51_400_000 // 64 * 64
src = torch.randn(n, device="cuda"); dst = torch.empty(n, device="cuda", dtype=torch.bfloat16)
us = timeit(lambda: lib.check(L.mmf_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), n, lib.stream_ptr())))
print(f"cast f32->bf16 {n/1e6:.1f} M elements: {us:7.1f} us  {6.0*n/us/1e3:7.0f} GB/s")
us = timeit(lambda: lib.check(L.mmf_cast_bf16_to_f32(dst.data_ptr(), src.data_ptr(), n, lib.stream_ptr())))
print(f"cast bf16->f32 {n/1e6:.1f} M elements: {us:7.1f} us  {6.0*n/us/1e3:7.0f} GB/s")
d = 768
rows = [8192, 8192, 6400, 6400, 480, 480]
xs = [torch.randn(r, d, device="cuda").bfloat16().requires_grad_(True) for r in rows]
gs = [torch.nn.Parameter(torch.ones(d, device="cuda")) for _ in rows]; bs = [torch.nn.Parameter(torch.zeros(d, device="cuda")) for _ in rows]
for p in gs + bs: p.grad = torch.zeros_like(p)
nb = sum(rows) * d * 2
us = timeit(lambda: ops.layernorm_group([(x.detach(), g, b) for x, g, b in zip(xs, gs, bs)]))
print(f"layernorm fwd x6 ({sum(rows)} rows): {us:7.1f} us  {2.0*nb/us/1e3:7.0f} GB/s")
ys = ops.layernorm_group([(x, g, b) for x, g, b in zip(xs, gs, bs)])
dys = [torch.randn_like(y) for y in ys]
torch.autograd.backward(ys, dys, retain_graph=True, inputs=xs)       # eager (the backward allocates its workspace)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): torch.autograd.backward(ys, dys, retain_graph=True, inputs=xs)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"layernorm bwd x6 (eager, incl. launch gaps): {us:7.1f} us  {3.0*nb/us/1e3:7.0f} GB/s")
a3 = [torch.randn(8192, d, device="cuda").bfloat16() for _ in range(3)]
us = timeit(lambda: ops.add3(*a3))
print(f"add3 8192x768: {us:7.1f} us  {4.0*8192*d*2/us/1e3:7.0f} GB/s")
mp = [torch.randn(16, t, d, device="cuda").bfloat16() for t in (512, 400, 30)]
us = timeit(lambda: ops.meanpool_cat(mp))
print(f"meanpool_cat 16x(512,400,30)x768: {us:7.1f} us  {sum(m.numel() for m in mp)*2/us/1e3:7.0f} GB/s")
