# A/B of the GEMM tile->XCD mapping and wgrad problem order (one box, interleaved).
cd $GRAFT_REPO_ROOT
O=gpurun_out/abx; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
for r in 1 2; do
  MMF_GEMM_XCD_GRANULE=0 MMF_WGRAD_SORT=size timeout -k 10 200 $B > $O/legacy_$r.log 2>&1 &&
  timeout -k 10 200 $B > $O/g32_$r.log 2>&1 &&
  MMF_GEMM_XCD_GRANULE=16 timeout -k 10 200 $B > $O/g16_$r.log 2>&1 &&
  MMF_GEMM_XCD_GRANULE=8 timeout -k 10 200 $B > $O/g8_$r.log 2>&1 &&
  MMF_GEMM_XCD_GRANULE=0 timeout -k 10 200 $B > $O/g0k_$r.log 2>&1 || exit 1
done
for f in $O/*_?.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"achieved": [0-9.]*' $f | head -1); done
