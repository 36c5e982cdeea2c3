# Two gloo ranks sharing the one GPU of a box: the overlapped exchange (three graphs, range all-reduces) against the
# one-shot exchange, same inputs — checks the schedule's logic and the gradient checksum; says nothing about RCCL timing.
cd $GRAFT_REPO_ROOT
O=gpurun_out/dpr; rm -rf $O; mkdir -p $O
export MMF_BENCH_CHECKSUM=1
R="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port"
for w in mult train; do
  MMF_DP_OVERLAP=1 timeout -k 10 400 $R 29611 bench.py --gpus 2 --backend gloo --workload $w --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $O/${w}_overlap.log 2>&1 &&
  MMF_DP_OVERLAP=0 timeout -k 10 400 $R 29612 bench.py --gpus 2 --backend gloo --workload $w --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $O/${w}_oneshot.log 2>&1 || { tail -30 $O/${w}_*.log; exit 1; }
done
for f in $O/*.log; do echo $f; grep -o '"ms_per_step": [0-9.]*\|"grad_checksum": \[[^]]*\]\|"allreduce_overlaps_wgrad": [a-z]*' $f | tr '\n' ' '; echo; done
