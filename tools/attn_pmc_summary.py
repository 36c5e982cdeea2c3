#!/usr/bin/env python3
"""Per-launch averages of the counters tools/attn_pmc.sh collected, one column per batch size (workgroups per CU)."""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
tab = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "b*_s*"))):
    if not os.path.isdir(d):
        continue
    b = os.path.basename(d).split("_")[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            m = re.search(r"attn_\w+", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(0)
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        # a dispatch reports one row per counter (already summed over the chip) -> average over dispatches
        ndisp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            m = re.search(r"attn_\w+", r["Kernel_Name"])
            if m:
                ndisp[m.group(0)].add(r["Dispatch_Id"])
        for (k, c), (s, n) in acc.items():
            tab[(k.split("::")[-1][:40], c)][b] = s / max(1, len(ndisp[k]))
bs = sorted({b for v in tab.values() for b in v}, key=lambda x: int(x[1:]))
print(f"{'kernel / counter':72s}" + "".join(f"{b:>14s}" for b in bs))
for (k, c) in sorted(tab):
    print(f"{k + ' ' + c:72s}" + "".join(f"{tab[(k, c)].get(b, float('nan')):14.4g}" for b in bs))
