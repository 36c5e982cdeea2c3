# as dp_rehearsal.sh, MulT only, three wgrad parts / arena ranges
cd $GRAFT_REPO_ROOT
O=gpurun_out/dpr3; rm -rf $O; mkdir -p $O
export MMF_BENCH_CHECKSUM=1
R="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port"
MMF_DP_PARTS=3 MMF_DP_OVERLAP=1 timeout -k 10 400 $R 29631 bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $O/parts3.log 2>&1 || { tail -30 $O/parts3.log; exit 1; }
grep -o '"grad_checksum": \[[^]]*\]\|"allreduce_overlaps_wgrad": [a-z]*' $O/parts3.log | tr '\n' ' '; echo
