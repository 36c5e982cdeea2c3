#!/usr/bin/env python3
"""Per-kernel wave-stall / LDS / L2 summary of the passes tools/pmc_stall.sh collected.

    python tools/pmc_stall_summary.py gpurun_out/r01s profiles/<round>_pmc_stalls.txt

Columns (averages over the launches of a kernel symbol, SQ counters summed over the chip):
  wait      SQ_WAIT_ANY / SQ_WAVE_CYCLES          waves parked in s_waitcnt / s_barrier
  istall    SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     issue stalls (MFMA RAW, busy pipe)
  ilds      SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES     LDS issue stalls (a sub-bucket of istall)
  active    SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  mfma      SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CYCLES-equivalent): see pmc_summary.py for the GRBM form
  ldsconf   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2hit     TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import load, short  # noqa: E402


def avg(d, k, c):
    v = d.get(k, {}).get(c)
    return v[0] / v[1] if v and v[1] else None


def ratio(a, b):
    return None if a is None or not b else a / b


def fmt(x, p=True):
    if x is None:
        return "    -"
    return f"{100 * x:5.1f}" if p else f"{x:9.3g}"


def main():
    root, out = sys.argv[1:3]
    S1, S2, T = load(os.path.join(root, "sq1")), load(os.path.join(root, "sq2")), load(os.path.join(root, "tcc"))
    rows = []
    for k in sorted(set(S1) | set(S2) | set(T)):
        wc = avg(S1, k, "SQ_WAVE_CYCLES")
        if not wc or wc < 1e5:
            continue
        n = S1[k]["SQ_WAVE_CYCLES"][1]
        hit, miss = avg(T, k, "TCC_HIT_sum"), avg(T, k, "TCC_MISS_sum")
        rows.append((wc * n, short(k)[:70], n,
                     ratio(avg(S1, k, "SQ_WAIT_ANY"), wc), ratio(avg(S1, k, "SQ_WAIT_INST_ANY"), wc),
                     ratio(avg(S1, k, "SQ_WAIT_INST_LDS"), wc), ratio(avg(S1, k, "SQ_ACTIVE_INST_ANY"), wc),
                     ratio(avg(S1, k, "SQ_ACTIVE_INST_LDS"), wc),
                     ratio(avg(S2, k, "SQ_LDS_BANK_CONFLICT"), avg(S2, k, "SQ_LDS_IDX_ACTIVE")),
                     ratio(hit, (hit or 0) + (miss or 0)),
                     avg(S2, k, "SQ_INSTS_LDS"), avg(S2, k, "SQ_INSTS_VALU"), avg(S2, k, "SQ_WAVES")))
    rows.sort(reverse=True)
    lines = ["# rocprofv3 --pmc, eager MulT fwd+bwd (tools/pmc_stall.sh); % of SQ_WAVE_CYCLES unless noted",
             f"{'kernel':70s} {'n':>4s}  wait istall  ilds active actlds ldsconf l2hit   insts_lds insts_valu     waves"]
    for r in rows[:24]:
        lines.append(f"{r[1]:70s} {r[2]:4d} {fmt(r[3])} {fmt(r[4])} {fmt(r[5])} {fmt(r[6])} {fmt(r[7])}  {fmt(r[8])} {fmt(r[9])}  "
                     f"{fmt(r[10], False)} {fmt(r[11], False)} {fmt(r[12], False)}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
