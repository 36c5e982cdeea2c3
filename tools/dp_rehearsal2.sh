cd $GRAFT_REPO_ROOT
O=gpurun_out/dpr2; rm -rf $O; mkdir -p $O
export MMF_BENCH_CHECKSUM=1
R="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port"
i=0
for m in 0 0 1 1; do
  i=$((i+1))
  MMF_DP_OVERLAP=$m timeout -k 10 400 $R $((29620+i)) bench.py --gpus 2 --backend gloo --workload train --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $O/train_m${m}_$i.log 2>&1 || { tail -30 $O/train_m${m}_$i.log; exit 1; }
done
for f in $O/*.log; do echo $f; grep -o '"grad_checksum": \[[^]]*\]\|"allreduce_overlaps_wgrad": [a-z]*' $f | tr '\n' ' '; echo; done
