#!/usr/bin/env python3
"""Copy the summaries tools/prof_all.sh wrote under gpurun_out/<tag>/ into profiles/<tag>_* (the judged copies).

    python tools/collect_profiles.py r01b "note for the csv headers"
"""
import csv
import glob
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return line
    raise SystemExit(f"no bench line in {path}")


def stats_csv(src_dir, dst, cmd, note):
    files = glob.glob(os.path.join(src_dir, "**", "*kernel_stats.csv"), recursive=True)
    if len(files) != 1:
        raise SystemExit(f"{src_dir}: expected one kernel_stats.csv, found {files}")
    rows = list(csv.DictReader(open(files[0])))
    with open(dst, "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}   ({note})\n")
        fh.write("# kernels under 0.05 % omitted; kernels of concurrent streams overlap in time, their durations include the sharing\n")
        fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        w = csv.writer(fh)
        for r in rows:
            if float(r["Percentage"]) >= 0.05:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])


def main():
    tag = sys.argv[1]
    note = sys.argv[2] if len(sys.argv) > 2 else ""
    src, dst = os.path.join(REPO, "gpurun_out", tag), os.path.join(REPO, "profiles")
    p = lambda name: os.path.join(dst, f"{tag}_{name}")
    for log, name in (("bench_plain.log", "bench_line.json"), ("bench_dropout.log", "bench_line_dropout.json"),
                      ("bench_stats.log", "bench_line_under_rocprof.json"), ("bench_hier.log", "bench_line_hier.json"),
                      ("bench_train.log", "bench_line_train.json"), ("bench_hier_stats.log", "bench_line_hier_under_rocprof.json"),
                      ("bench_train_stats.log", "bench_line_train_under_rocprof.json"), ("bench_meld.log", "bench_line_meld.json"),
                      ("bench_meld_stats.log", "bench_line_meld_under_rocprof.json")):
        if os.path.exists(os.path.join(src, log)):
            open(p(name), "w").write(last_json_line(os.path.join(src, log)) + "\n")
    stats_csv(os.path.join(src, "stats"), p("bench_kernel_stats.csv"), "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline", "MulT fwd+bwd B=16 T=512/400/30 d=768; " + note)
    stats_csv(os.path.join(src, "stats_hier"), p("hier_kernel_stats.csv"), "python3 bench.py --workload hier --steps 20 --warmup 5 --no-cpu-baseline", "hierarchical fusion, sequence inputs; " + note)
    stats_csv(os.path.join(src, "stats_train"), p("train_kernel_stats.csv"), "python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline", "full training step; " + note)
    if os.path.isdir(os.path.join(src, "stats_meld")):
        stats_csv(os.path.join(src, "stats_meld"), p("meld_kernel_stats.csv"), "python3 bench.py --workload meld --steps 20 --warmup 5 --no-cpu-baseline", "MELD-shaped training step (BASELINE configs[4]); " + note)
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), p("pmc_traffic.json"),
                           os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), os.path.join(src, "pmc_mfma")])
    for log, name in (("attn_bench.log", "attention_generations.txt"), ("step_launches.log", "step_launches.txt"),
                      ("step_timeline.txt", "step_timeline.txt"), ("step_timeline_hier.txt", "step_timeline_hier.txt"),
                      ("step_timeline_train.txt", "step_timeline_train.txt"), ("step_timeline_meld.txt", "step_timeline_meld.txt"),
                      ("gemm7_vs_gemm6.log", "gemm7_vs_gemm6.txt"), ("hipblaslt_yardstick.log", "hipblaslt_yardstick.txt"), ("parity_bf16.txt", "parity_vs_bf16_storage_oracle.txt"),
                      ("parity_fp32.txt", "parity_vs_fp32_oracle.txt")):
        if os.path.exists(os.path.join(src, log)):
            open(p(name), "w").write(open(os.path.join(src, log)).read())


if __name__ == "__main__":
    main()
