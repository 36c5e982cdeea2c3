#!/bin/bash
# Per-KERNEL durations of the attention launches of chosen tools/attn_bench.py cases (plain launches under rocprofv3 --kernel-trace):
#   CASES="stepAV,a<-t" bash tools/attn_kernel_times.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/${TAG:-attn_kt}; mkdir -p $O
for nar in 0; do
  for c in ${CASES//,/ }; do
    d=$O/trace_${c//[<>-]/_}_$nar
    MMF_ATTN_NARROW=$nar MMF_ATTN_NOGRAPH=1 MMF_ATTN_CASES="$c" timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python3 tools/attn_bench.py ${WHAT:-bwd} > $O/log_${c//[<>-]/_}_$nar.txt 2>&1
    echo "== case $c narrow=$nar"
    python3 - "$d" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    n = row["Name"]
    if "attn" in n:
        print(f"  {n[:70]:70s} calls {row['Calls']:>4s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
    rm -rf $d
  done
done
