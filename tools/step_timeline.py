#!/usr/bin/env python3
"""Per-kernel timeline of ONE replayed step from a rocprofv3 --kernel-trace csv (tools/prof_all.sh writes it).

    python tools/step_timeline.py gpurun_out/r01b/stats profiles/r01b_step_timeline.txt [step_index]

A step starts at the gradient arena's hole-zeroing launch (zero_ranges_kernel, the first kernel of arena.zero_grad(lazy=True));
the 21st such step of the trace is a timed graph replay.
Columns: start (us from the step's first kernel), duration (us), hardware queue, kernel."""
import collections
import csv
import glob
import os
import sys


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:72]


def main():
    src, out = sys.argv[1:3]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "zero_ranges_kernel" in r["Kernel_Name"]]
    if len(starts) <= k + 1:                    # older traces: the whole-arena weight cast opened every step
        starts = [i for i, r in enumerate(rows) if "cast_f32_bf16_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 200000]
    step = rows[starts[k]:starts[k + 1]]
    t0 = int(step[0]["Start_Timestamp"])
    ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Queue_Id"], short(r["Kernel_Name"])) for r in step]
    pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
    lvl, last, hist = 0, 0, collections.Counter()
    for t, d in pts:
        hist[lvl] += t - last
        last, lvl = t, lvl + d
    lines = [f"# one replayed step of `bench.py` (step {k} of the trace in {src}); span {max(e for _, e, _, _ in ev) / 1e3:.1f} us, "
             f"{len(ev)} kernels; time with 0 / 1 / 2+ kernels running: {hist[0] / 1e3:.0f} / {hist[1] / 1e3:.0f} / "
             f"{sum(v for c, v in hist.items() if c >= 2) / 1e3:.0f} us",
             "# start_us  dur_us  queue  kernel"]
    lines += [f"{s / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  {n}" for s, e, q, n in ev]
    open(out, "w").write("\n".join(lines) + "\n")
    print(lines[0])


if __name__ == "__main__":
    main()
