#!/usr/bin/env python3
"""Attention microbench on the MulT shapes (six cross problems / three self problems in one launch)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import ops
B, H, dh, d = 16, 8, 96, 768
T = dict(t=512, a=400, v=30)
def run(pairs, bwd, reps=20):
    srcs, specs = [], []
    for i, (q, k) in enumerate(pairs):
        srcs.append(torch.randn(B * T[q], d, device="cuda").bfloat16().requires_grad_(bwd))
        srcs.append(torch.randn(B * T[k], 2 * d, device="cuda").bfloat16().requires_grad_(bwd))
        specs.append(ops.AttnSpec(B, T[q], T[k], q=(2 * i, 0), k=(2 * i + 1, 0), v=(2 * i + 1, d)))
    gos = None
    def once():
        outs = ops.attention_group(specs, H, dh, srcs)
        if bwd:
            torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): once()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()            # replay: no host launch overhead in the measurement
    with torch.cuda.graph(g):
        once()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = sum(4.0 * B * H * T[q] * T[k] * dh for q, k in pairs) * (3 if bwd else 1)
    return us, fl / us / 1e6
cross = [("t", "a"), ("t", "v"), ("a", "t"), ("a", "v"), ("v", "t"), ("v", "a")]
selfp = [("t", "t"), ("a", "a"), ("v", "v")]
tag = os.environ.get("MMF_ATTN_DEBUG", "0")
for name, pairs in [("cross x6", cross), ("self x3", selfp), ("t<-a only", [("t", "a")]), ("t<-t only", [("t", "t")])]:
    us, tf = run(pairs, False)
    print(f"dbg{tag} fwd {name:10s} {us:8.1f} us {tf:7.1f} TF", flush=True)
if tag == "0":
    for name, pairs in [("cross x6", cross), ("self x3", selfp)]:
        us, tf = run(pairs, True)
        print(f"dbg{tag} fwd+bwd {name:10s} {us:8.1f} us {tf:7.1f} TF(3x fwd flops)", flush=True)
