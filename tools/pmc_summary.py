#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into profiles/<round>_pmc_traffic.json.

    python tools/pmc_summary.py OUT.json FETCH_DIR WRITE_DIR [MFMA_DIR]

FETCH_DIR / WRITE_DIR / MFMA_DIR are the -d directories of three separate runs of the same command
(`rocprofv3 --kernel-trace --pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`;
one counter set per run, as MI355X_MICROARCH.md prescribes).  Per kernel symbol, averaged over its launches:
  hbm_bytes_per_launch = 2 * FETCH_SIZE (KiB; gfx950 tallies 128-B requests of a wide streaming read at 64 B) + WRITE_SIZE (KiB)
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * GRBM_GUI_ACTIVE / 8)   (GRBM_GUI_ACTIVE sums the 8 XCDs)
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


def short(name):
    for pre in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    cut = name.find("((anonymous")
    if cut < 0:
        cut = name.find("(")
    return name[:cut] if cut > 0 and not name.startswith("at::") else name[:90]


def main():
    out, fdir, wdir = sys.argv[1:4]
    mdir = sys.argv[4] if len(sys.argv) > 4 else None
    F, Wr = load(fdir), load(wdir)
    M = load(mdir) if mdir else {}
    kernels = {}
    for k in sorted(set(F) | set(Wr)):
        f, w = F.get(k, {}).get("FETCH_SIZE"), Wr.get(k, {}).get("WRITE_SIZE")
        if not f and not w:
            continue
        fetch = 2.0 * 1024.0 * f[0] / f[1] if f else 0.0
        write = 1024.0 * w[0] / w[1] if w else 0.0
        e = {"launches": int((f or w)[1]), "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
             "hbm_bytes_per_launch": int(fetch + write)}
        m = M.get(k, {})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"][0] > 0:
            e["mfma_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * m["GRBM_GUI_ACTIVE"][0] / 8.0), 4)
        kernels[short(k)] = e
    doc = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                   "(three separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-graph --profile-steps 1 --no-cpu-baseline "
                   "(MulT fwd+bwd, eager); FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                   "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); averages over all launches "
                   "of the kernel symbol (tools/pmc_summary.py)",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, e in kernels.items():
        if e["hbm_bytes_per_launch"] > 5e6:
            print(f"{k[:60]:60s} {e['launches']:4d}x  {e['hbm_bytes_per_launch'] / 1e6:8.1f} MB/launch  mfma_util {e.get('mfma_util')}")


if __name__ == "__main__":
    main()
