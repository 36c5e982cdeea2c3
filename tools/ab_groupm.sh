cd $GRAFT_REPO_ROOT
O=gpurun_out/abm; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
for r in 1 2; do
  for g in 8 4 2 16; do
    MMF_GEMM_GROUP_M=$g timeout -k 10 200 $B > $O/gm${g}_$r.log 2>&1 || exit 1
  done
done
for f in $O/*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"achieved": [0-9.]*' $f | head -1) $(grep -o 'gemm2_grouped_kernel<NN,bf16>": {"us_per_step": [0-9.]*' $f) $(grep -o 'gemm2_grouped_kernel<NT,bf16>": {"us_per_step": [0-9.]*' $f); done
