cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for b in 512 2048; do
  MMF_LN_BWD_BLOCKS=$b timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ln_$b --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ln_$b.log 2>&1
  echo "blocks $b: $(grep -h 'ln_bwd_kernel\|ln_fwd_kernel\|ln_bwd_finalize' gpurun_out/ln_$b/*/*kernel_stats.csv | cut -d, -f1,4 | sed 's/void (anonymous namespace):://; s/((anonymous.*)"//' | tr '\n' ' ')"
done
