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

# where do the copies come from?  log every .contiguous() / .clone() that really copies, with its call site
import collections, traceback
sites = collections.Counter()
_orig_contig, _orig_clone = torch.Tensor.contiguous, torch.Tensor.clone
def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mmfusion" in fr.filename or "models" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}"
    return "?"
def contig(self, *a, **k):
    if self.is_cuda and not self.is_contiguous():
        sites["contiguous " + str(tuple(self.shape)) + " " + _site()] += 1
    return _orig_contig(self, *a, **k)
def clone(self, *a, **k):
    if self.is_cuda:
        sites["clone " + str(tuple(self.shape)) + " " + _site()] += 1
    return _orig_clone(self, *a, **k)
torch.Tensor.contiguous, torch.Tensor.clone = contig, clone
step(); torch.cuda.synchronize()
torch.Tensor.contiguous, torch.Tensor.clone = _orig_contig, _orig_clone
for k, v in sites.most_common(30):
    print(f"x{v}  {k[:170]}")
