#!/usr/bin/env python3
"""Which aten ops (torch glue) launch kernels in one eager step: op name, calls, CUDA time.  python tools/aten_ops.py [mult|hier]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
os.environ.setdefault("MMFUSION_CONFIG_MKDIRS", "0")
import torch, bench
from torch.profiler import profile, ProfilerActivity
from mmfusion import arena as arena_mod
wl = sys.argv[1] if len(sys.argv) > 1 else "mult"
cfg, model, xs = bench.build(wl, torch.device("cuda", 0), 0)
arena = arena_mod.ensure(model)
step = bench.make_step(wl, model, xs, arena)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:45]:
    stack = " <- ".join(s.split("/")[-1] for s in e.stack[:3]) if e.stack else ""
    print(f"{e.key:32s} x{e.count:3d} {e.device_time_total:8.1f} us   {stack[:150]}")
