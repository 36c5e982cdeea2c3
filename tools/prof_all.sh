# Round profile collection on the GPU box:  TAG=r04 bash tools/prof_all.sh   (then: python tools/collect_profiles.py r04 "note")
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
TAG=${TAG:-r04}
O=gpurun_out/$TAG
mkdir -p $O
# PART=a: the rocprofv3 passes; PART=b: bench lines, microbenchmarks, parity reports (two gpurun calls of <= 20 minutes); default: both
if [ "${PART:-ab}" != "b" ]; then
timeout -k 10 200 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1 || echo SMOKE_FAILED
export MMF_BENCH_NO_FROZEN=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_stats.log 2>&1
python3 tools/step_timeline.py $O/stats $O/step_timeline.txt
# PMC passes with the stream concurrency off (one launch per stage, as in bench.py's per-kernel timing pass)
export MMF_MULT_STREAMS=1 MMF_HIER_STREAMS=0
PM="python3 bench.py --steps 2 --warmup 1 --no-graph --profile-steps 1 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $PM > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $PM > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
unset MMF_MULT_STREAMS MMF_HIER_STREAMS
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_hier --output-format csv -- python3 bench.py --workload hier --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_hier_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_train --output-format csv -- python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_meld --output-format csv -- python3 bench.py --workload meld --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_meld_stats.log 2>&1
python3 tools/step_timeline.py $O/stats_hier $O/step_timeline_hier.txt 18
python3 tools/step_timeline.py $O/stats_train $O/step_timeline_train.txt 18
python3 tools/step_timeline.py $O/stats_meld $O/step_timeline_meld.txt 18
rm -f $O/stats_hier/*/*kernel_trace.csv $O/stats_train/*/*kernel_trace.csv $O/stats_meld/*/*kernel_trace.csv
fi
if [ "${PART:-ab}" != "a" ]; then
export MMF_BENCH_NO_FROZEN=1
MMF_ATTN_IMPLS=2 timeout -k 10 200 python3 tools/attn_bench.py both > $O/attn_bench.log 2>&1
timeout -k 10 200 python3 tools/step_launches.py mult > $O/step_launches.log 2>&1
unset MMF_BENCH_NO_FROZEN
python3 bench.py > $O/bench_plain.log 2>&1
python3 bench.py --dropout 0.1 --no-cpu-baseline > $O/bench_dropout.log 2>&1
python3 bench.py --workload hier --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_hier.log 2>&1
python3 bench.py --workload train --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_train.log 2>&1
python3 bench.py --workload meld --steps 100 --warmup 20 > $O/bench_meld.log 2>&1
timeout -k 10 300 python3 tools/gemm7_bench.py > $O/gemm7_vs_gemm6.log 2>&1
timeout -k 10 300 python3 tools/blaslt_yardstick.py > $O/hipblaslt_yardstick.log 2>&1
ORACLE_STORAGE=bf16 TOPK=4 python3 tools/parity_report.py > $O/parity_bf16.txt 2>&1
TOPK=4 python3 tools/parity_report.py > $O/parity_fp32.txt 2>&1
fi
ls $O
