# Two gloo ranks sharing the one GPU of a box: the overlapped exchange (three graphs, range all-reduces) against the
# one-shot exchange, same inputs — checks the schedule's logic and the gradient checksum; says nothing about RCCL timing.
# Round 2: the ranks are started by bench.py ITSELF (`python bench.py --gpus 2`: a torch.distributed.run child process),
# and every rank leaves the process group before rank 0's per-kernel pass, so this also rehearses the launch path the
# driver's scaling run may take and the control flow that replaced the two round-1 hangs.
cd $GRAFT_REPO_ROOT
O=gpurun_out/dpr; rm -rf $O; mkdir -p $O
export MMF_BENCH_CHECKSUM=1
for w in mult train; do
  MMF_DP_OVERLAP=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --workload $w --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $O/${w}_overlap.log 2>&1 &&
  MMF_DP_OVERLAP=0 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --workload $w --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $O/${w}_oneshot.log 2>&1 || { tail -30 $O/${w}_*.log; exit 1; }
done
for f in $O/*.log; do echo $f; grep -o '"n_gpus": [0-9]*\|"ms_per_step": [0-9.]*\|"grad_checksum": \[[^]]*\]\|"allreduce_overlaps_wgrad": [a-z]*\|"overlap_fallback": [^,]*' $f | tr '\n' ' '; echo; done
