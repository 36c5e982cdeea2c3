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

# ---- the step's own launches: every grouped launch of the MulT step (profiles/r03_step_launches.txt) as ONE mmf launch next to
# the same problems as back-to-back torch.matmul calls (hipBLASLt picks a kernel per problem), both in a captured graph.
STEP = [
    ("NT in-proj x8", "NT", [(6400, 768, 768)] * 2 + [(8192, 1536, 768)] * 2 + [(480, 1536, 768)] + [(480, 768, 768)] * 2 + [(6400, 1536, 768)]),
    ("NT out-proj a,v", "NT", [(6400, 768, 768)] * 2 + [(480, 768, 768)] * 2),
    ("NT ffn1 a,v", "NT", [(6400, 3072, 768)] * 2 + [(480, 3072, 768)] * 2),
    ("NT ffn2 a,v", "NT", [(6400, 768, 3072)] * 2 + [(480, 768, 3072)] * 2),
    ("NT self qkv a,v", "NT", [(6400, 2304, 768), (480, 2304, 768)]),
    ("NT out-proj t", "NT", [(8192, 768, 768)] * 2),
    ("NT ffn1 t", "NT", [(8192, 3072, 768)] * 2),
    ("NT ffn2 t", "NT", [(8192, 768, 3072)] * 2),
    ("NT self qkv t", "NT", [(8192, 2304, 768)]),
    ("NN self qkv t", "NN", [(8192, 768, 2304)]),
    ("NN ffn2 dgrad t", "NN", [(8192, 3072, 768)] * 2),
    ("NN ffn1 dgrad t", "NN", [(8192, 768, 3072)] * 2),
    ("NN out-proj dgrad t", "NN", [(8192, 768, 768)] * 2),
    ("NN self qkv a,v", "NN", [(6400, 768, 2304), (480, 768, 2304)]),
    ("NN ffn2 dgrad a,v", "NN", [(6400, 3072, 768)] * 2 + [(480, 3072, 768)] * 2),
    ("NN ffn1 dgrad a,v", "NN", [(6400, 768, 3072)] * 2 + [(480, 768, 3072)] * 2),
    ("NN out-proj dgrad a,v", "NN", [(6400, 768, 768)] * 2 + [(480, 768, 768)] * 2),
    ("NN in-proj dgrad x8", "NN", [(6400, 768, 768)] * 2 + [(8192, 768, 1536)] * 2 + [(480, 768, 1536)] + [(480, 768, 768)] * 2 + [(6400, 768, 1536)]),
    ("TN all weight gradients", "TN",
     sum(([(768, 3072, T)] * 2 + [(3072, 768, T)] * 2 + [(2304, 768, T)] + [(1536, 768, T)] * 2 + [(768, 768, T)] * 4
          for T in (8192, 6400, 480)), [])),
]
tot = {"NT": [0.0, 0.0, 0.0], "NN": [0.0, 0.0, 0.0], "TN": [0.0, 0.0, 0.0]}
print("\n# the step's grouped launches: (M, N, K) = output rows, output columns, reduction; hipBLASLt = the same problems as separate torch.matmul calls")
for name, kind, probs in STEP:
    ts, ms, fl = [], [], 0.0
    for (M, N, K) in probs:
        fl += 2.0 * M * N * K / 1e6
        if kind == "NT":      # y[M][N] = x[M][K] w[N][K]^T
            x, w = torch.randn(M, K, device="cuda").bfloat16(), torch.randn(N, K, device="cuda").bfloat16()
            y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ts.append((x, w.t(), y)); ms.append((x, w, y, None, None))
        elif kind == "NN":    # dx[M][N] = dy[M][K] w[K][N]
            g, w = torch.randn(M, K, device="cuda").bfloat16(), torch.randn(K, N, device="cuda").bfloat16()
            y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ts.append((g, w, y)); ms.append((g, w, y, None, None))
        else:                 # dw[M][N] = dy[K][M]^T x[K][N]   (f32 out for mmf, bf16 for torch: the lighter job)
            g, x = torch.randn(K, M, device="cuda").bfloat16(), torch.randn(K, N, device="cuda").bfloat16()
            y32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
            y16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ts.append((g.t(), x, y16)); ms.append((g, x, y32, None, None))
    code = {"NT": lib.GEMM_NT, "NN": lib.GEMM_NN, "TN": lib.GEMM_TN}[kind]

    def run_lt():
        for a, b, o in ts:
            torch.matmul(a, b, out=o)
    t_lt = timeit(run_lt, reps=10, inner=3)
    t_m = timeit(lambda: ops.gemm_group(code, ms, 0), reps=10, inner=3)
    tot[kind][0] += t_lt; tot[kind][1] += t_m; tot[kind][2] += fl
    print(f"{name:26s} {len(probs):2d} problems  hipBLASLt {t_lt:7.1f} us {fl / t_lt:7.1f} TF   mmf grouped {t_m:7.1f} us {fl / t_m:7.1f} TF", flush=True)
for k, (a, b, f) in tot.items():
    print(f"sum {k}: hipBLASLt {a:7.1f} us ({f / a:6.1f} TF)   mmf {b:7.1f} us ({f / b:6.1f} TF)")
