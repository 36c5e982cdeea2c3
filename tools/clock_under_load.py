import os, sys, subprocess, threading, time
sys.path.insert(0, "simple-multimodal_amd"); sys.path.insert(0, "tools")
import torch
from mmfusion import lib, ops
from mmfusion.lib import GEMM_NT
L = lib.load()
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l]
            samples.append((time.time(), " | ".join(x.split(":", 1)[-1].strip() if False else x.strip() for x in s)))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.3)
th = threading.Thread(target=poll); th.start()
time.sleep(1.5)
M = N = K = 4096
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16() * K ** -0.5
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 0
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.gemm_group(GEMM_NT, [(A, B, C, None, None)], 0)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
print("gemm 4096^3 x", n, "avg us", e0.elapsed_time(e1) * 1e3 / n, "TF", 2 * M * N * K * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
time.sleep(1.0)
stop = True; th.join()
for t, s in samples: print(f"{t - t0:6.2f}s  {s[:300]}")
