cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01c
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_hier --output-format csv -- python3 bench.py --workload hier --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_hier_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_train --output-format csv -- python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train_stats.log 2>&1
python3 bench.py --workload hier --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_hier.log 2>&1
python3 bench.py --workload train --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_train.log 2>&1
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_mult.log 2>&1
