#!/usr/bin/env python3
"""Print a per-tensor parity table (HIP path vs CPU oracle) for golden cases.  GPU box only.
    python tools/parity_report.py [case ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "simple-multimodal_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("MMFUSION_CONFIG_MKDIRS", "0")

from golden_cases import CASES                      # noqa: E402
from helpers import l2_rel, load_fixture, oracle_fwd_bwd, rel_err   # noqa: E402
from test_parity_gpu import run_hip                 # noqa: E402


def main():
    for name in (sys.argv[1:] or sorted(CASES)):
        fx = load_fixture(name)
        out, gin, gp = run_hip(fx.meta)
        storage = os.environ.get("ORACLE_STORAGE", "fp32")      # bf16: the oracle rounds at the HIP path's storage points
        o_out, o_gin, o_gp = oracle_fwd_bwd(fx.meta, storage=storage)
        print(f"== {name} (oracle storage {storage})")
        for k in sorted(out):
            print(f"   out  {k:45s} abs {float((out[k] - o_out[k]).abs().max()):.3e} rel {rel_err(out[k], o_out[k]):.3e}")
        for i, g in enumerate(gin):
            print(f"   gin  {i:<45d} max {rel_err(g, o_gin[i]):.3e} l2 {l2_rel(g, o_gin[i]):.3e}")
        rows = sorted(((l2_rel(gp[k], o_gp[k]), k) for k in gp), reverse=True)
        for e, k in rows[:int(os.environ.get("TOPK", "12"))]:
            print(f"   gpar {k:45s} l2 {e:.3e} max {rel_err(gp[k], o_gp[k]):.3e} norm ratio "
                  f"{float(gp[k].norm() / (o_gp[k].norm() + 1e-30)):.4f}")


if __name__ == "__main__":
    main()
