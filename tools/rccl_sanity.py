#!/usr/bin/env python3
"""One-rank RCCL sanity check on the GPU box: init the nccl backend the way bench.py does, run bucketed async
all-reduces of a bf16 wire buffer and the widen kernel.  (Two ranks cannot share one GPU under RCCL, so the N > 1
arithmetic is covered by the gloo tests; this only proves the RCCL path loads and runs in this environment.)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "simple-multimodal_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from mmfusion import dp
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.randn(60_000_000, device=dev).bfloat16()
ref = x.clone()
works = [dist.all_reduce(x[s:e], op=dist.ReduceOp.SUM, async_op=True) for s, e in dp.bucket_bounds(x.numel(), 2)]
for w in works: w.wait()
torch.cuda.synchronize()
assert torch.equal(x, ref)
dist.barrier()
t = torch.tensor([1.0], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.destroy_process_group()
print("rccl one-rank sanity ok:", len(works), "buckets")
