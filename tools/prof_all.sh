set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r01b
mkdir -p $O
timeout -k 10 120 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1 || echo SMOKE_FAILED
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 > $O/bench_stats.log 2>&1
# PMC passes with the stream concurrency off (one launch per stage, as in bench.py's per-kernel timing pass)
export MMF_MULT_STREAMS=1 MMF_HIER_STREAMS=0
PM="python3 bench.py --steps 2 --warmup 1 --no-graph --profile-steps 1 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $PM > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $PM > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
unset MMF_MULT_STREAMS MMF_HIER_STREAMS
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_hier --output-format csv -- python3 bench.py --workload hier --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_hier_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_train --output-format csv -- python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train_stats.log 2>&1
timeout -k 10 200 python3 tools/attn_bench.py both > $O/attn_bench.log 2>&1
timeout -k 10 200 python3 tools/step_launches.py mult > $O/step_launches.log 2>&1
python3 bench.py --steps 50 --warmup 10 > $O/bench_plain.log 2>&1
python3 bench.py --steps 50 --warmup 10 --dropout 0.1 --no-cpu-baseline > $O/bench_dropout.log 2>&1
ls $O
