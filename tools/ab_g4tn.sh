cd $GRAFT_REPO_ROOT
O=gpurun_out/abg; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad or lazy" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 200 $B > $O/auto_1.log 2>&1 &&
MMF_GEMM_IMPL=4 timeout -k 10 200 $B > $O/g4_1.log 2>&1 &&
timeout -k 10 200 $B > $O/auto_2.log 2>&1 &&
MMF_GEMM_IMPL=4 timeout -k 10 200 $B > $O/g4_2.log 2>&1 || exit 1
for f in $O/auto_?.log $O/g4_?.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); python3 - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        for k,v in d['kernels'].items(): print('   ',k,v)
PY
done
