#!/usr/bin/env python3
"""Attention microbench on the MulT shapes (six cross problems / three self problems in one launch).

Calls the C ABI directly (no autograd, no allocation in the timed region); each measurement is one hipGraph
holding INNER back-to-back launches, replayed REPS times, so launch gaps do not enter the number.
    python tools/attn_bench.py [fwd|bwd|both]       MMF_ATTN_IMPLS lists the implementations timed (2 = attention2.hip, the only one since round 3)
"""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib

B, H, dh, d = 16, 8, 96, 768
T = dict(t=int(os.environ.get("MMF_ATTN_TT", 512)), a=int(os.environ.get("MMF_ATTN_TA", 400)), v=30)
INNER, REPS = 10, 10


def problems(pairs, bwd):
    keep, arr = [], (lib.AttnProblem * len(pairs))()
    for i, (q, k) in enumerate(pairs):
        Q = torch.randn(B * T[q], d, device="cuda").bfloat16()
        KV = torch.randn(B * T[k], 2 * d, device="cuda").bfloat16()
        O = torch.empty_like(Q)
        LSE = torch.empty(B * H * T[q], device="cuda")
        dO, dQ, dKV = torch.randn_like(Q), torch.empty_like(Q), torch.empty_like(KV)
        delta = torch.empty_like(LSE)
        keep += [Q, KV, O, LSE, dO, dQ, dKV, delta]
        p = arr[i]
        p.Q, p.K, p.V, p.O, p.LSE = Q.data_ptr(), KV.data_ptr(), KV.data_ptr() + 2 * d, O.data_ptr(), LSE.data_ptr()
        p.dO, p.delta, p.dQ, p.dK, p.dV = dO.data_ptr(), delta.data_ptr(), dQ.data_ptr(), dKV.data_ptr(), dKV.data_ptr() + 2 * d
        p.B, p.H, p.Tq, p.Tk = B, H, T[q], T[k]
        p.ldq, p.ldk, p.ldv, p.ldo = d, 2 * d, 2 * d, d
    return arr, keep


def run(pairs, bwd):
    L = lib.load()
    arr, keep = problems(pairs, bwd)
    scale = dh ** -0.5
    fn = L.mmf_attn_bwd_grouped if bwd else L.mmf_attn_fwd_grouped

    def once():
        lib.check(fn(arr, len(pairs), dh, scale, lib.stream_ptr()))
    if bwd:                                   # LSE must be valid
        lib.check(L.mmf_attn_fwd_grouped(arr, len(pairs), dh, scale, lib.stream_ptr()))
    if os.environ.get("MMF_ATTN_NOGRAPH"):    # for rocprofv3 --pmc runs: plain launches, the profiler times / counts each
        for _ in range(10):
            once()
        torch.cuda.synchronize()
        return 0.0, 0.0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            once()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(INNER):
            once()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (REPS * INNER)
    fl = sum(4.0 * B * H * T[q] * T[k] * dh for q, k in pairs) * (2 if bwd else 1)
    return us, fl / us / 1e6


cross = [("t", "a"), ("t", "v"), ("a", "t"), ("a", "v"), ("v", "t"), ("v", "a")]
selfp = [("t", "t"), ("a", "a"), ("v", "v")]
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
impls = [int(x) for x in os.environ.get("MMF_ATTN_IMPLS", "2").split(",")]
cases = [("cross x6", cross), ("self x3", selfp), ("t<-a only", [("t", "a")]), ("a<-t only", [("a", "t")]),
         ("t<-t only", [("t", "t")]), ("v<-t only", [("v", "t")]), ("t<-v only", [("t", "v")]),
         ("big2 t<-a,a<-t", [("t", "a"), ("a", "t")]), ("narrow4 x30", [("t", "v"), ("a", "v"), ("v", "t"), ("v", "a")]),
         ("grpA t<-a,a<-v,v<-t", [("t", "a"), ("a", "v"), ("v", "t")]), ("grpB t<-v,a<-t,v<-a", [("t", "v"), ("a", "t"), ("v", "a")]),
         # the launches of the two-stream step (tools/step_launches.py)
         ("stepAV a<-t,a<-v,v<-t,v<-a", [("a", "t"), ("a", "v"), ("v", "t"), ("v", "a")]), ("stepT t<-a,t<-v", [("t", "a"), ("t", "v")]),
         ("selfAV a<-a,v<-v", [("a", "a"), ("v", "v")])]
if os.environ.get("MMF_ATTN_CASES"):
    want = os.environ["MMF_ATTN_CASES"].split(",")
    cases = [c for c in cases if c[0].split()[0] in want]
if os.environ.get("MMF_ATTN_B"):
    B = int(os.environ["MMF_ATTN_B"])
for bwd in ([False] if what == "fwd" else [True] if what == "bwd" else [False, True]):
    for name, pairs in cases:
        row = f"{'bwd' if bwd else 'fwd'} {name:20s}"
        for impl in impls:
            lib.check(lib.load().mmf_attn_select_impl(impl))
            us, tf = run(pairs, bwd)
            row += f"   impl{impl} {us:7.1f} us {tf:6.1f} TF"
        print(row + ("   (bwd credited 8 Tq Tk d)" if bwd else ""), flush=True)
lib.check(lib.load().mmf_attn_select_impl(0))
