cd $GRAFT_REPO_ROOT
O=gpurun_out/abx2; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for w in hier train; do
  B="python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline"
  for r in 1 2; do
    MMF_GEMM_XCD_GRANULE=0 MMF_WGRAD_SORT=size timeout -k 10 200 $B > $O/${w}_legacy_$r.log 2>&1 &&
    timeout -k 10 200 $B > $O/${w}_g32_$r.log 2>&1 || exit 1
  done
done
for f in $O/*_?.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"achieved": [0-9.]*' $f | head -1); done
