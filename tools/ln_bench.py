#!/usr/bin/env python3
"""LayerNorm forward / backward through the C ABI on the MulT row counts (hipGraph of 10 launches; algorithmic bytes:
forward 2 x 2d, backward 3 x 2d per row)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
import torch
from mmfusion import lib
from mmfusion.lib import LnProblem

d = 768
L = lib.load()


def timed(fn, inner=10, reps=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (inner * reps)


for name, rows in (("3 problems 8192/6400/480", [8192, 6400, 480]), ("text pair 8192 x2", [8192, 8192]),
                   ("a+v 6400 x2 + 480 x2", [6400, 6400, 480, 480]), ("all six", [8192, 8192, 6400, 6400, 480, 480])):
    keep, fw, bw = [], [], []
    for r in rows:
        x = torch.randn(r, d, device="cuda").bfloat16(); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
        g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda"); st = torch.empty(2, r, device="cuda")
        dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
        keep += [x, y, dy, dx, g, b, st, dg, db]
        fw.append(LnProblem(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), None, None, None, None, r))
        bw.append(LnProblem(x.data_ptr(), None, g.data_ptr(), None, st[0].data_ptr(), st[1].data_ptr(), dy.data_ptr(), dx.data_ptr(),
                            dg.data_ptr(), db.data_ptr(), r))
    ws = torch.empty(L.mmf_layernorm_bwd_workspace_bytes(d) // 4, device="cuda")
    nb = sum(rows) * d * 2
    lib.layernorm_fwd_grouped(fw, d, 1e-5)
    usf = timed(lambda: lib.layernorm_fwd_grouped(fw, d, 1e-5))
    usb = timed(lambda: lib.layernorm_bwd_grouped(bw, d, ws))
    print(f"{name:28s} fwd {usf:6.1f} us {2.0 * nb / usf / 1e3:6.0f} GB/s    bwd (+finalize) {usb:6.1f} us {3.0 * nb / usb / 1e3:6.0f} GB/s", flush=True)
