#!/usr/bin/env python3
"""Per-launch timing of one eager MulT / hier step: every grouped GEMM / attention launch in issue order with
its problem shapes, HIP-event duration and algorithmic TFLOP/s (median of REPS steps).
    python tools/step_launches.py [mult|hier|train]"""
import os
import sys
import statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
os.environ.setdefault("MMFUSION_CONFIG_MKDIRS", "0")
import torch
import bench
from mmfusion import arena as arena_mod, lib

workload = sys.argv[1] if len(sys.argv) > 1 else "mult"
REPS = 7
dev = torch.device("cuda", 0)
cfg, model, xs = bench.build(workload if workload != "train" else "hier", dev, 0)
arena = arena_mod.ensure(model)
step = bench.make_step(workload if workload != "train" else "hier", model, xs, arena)
for _ in range(3):
    step()
runs = []
for _ in range(REPS):
    lib.PROFILE = []
    step()
    torch.cuda.synchronize()
    runs.append([(l, f, e0.elapsed_time(e1) * 1e3, d) for l, f, e0, e1, d in lib.PROFILE])
lib.PROFILE = None
n = len(runs[0])
total = 0.0
for i in range(n):
    label, flops, _, detail = runs[0][i]
    us = statistics.median(r[i][2] for r in runs)
    total += us
    shapes = {}
    for s in detail or []:
        shapes[s] = shapes.get(s, 0) + 1
    desc = " ".join(f"{c}x{s}" for s, c in shapes.items())
    print(f"{i:3d} {label:32s} {us:8.1f} us {flops / us / 1e6 if us > 0 else 0:7.1f} TF  {desc}")
print(f"sum of timed launches {total:.1f} us")
