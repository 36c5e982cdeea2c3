# A/B of the TN inner loop (compile-time MMF_TN_PIPE) on one box: prebuilt = 1, rebuilt on the box = 0, then 1 again.
cd $GRAFT_REPO_ROOT
O=gpurun_out/abt; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
G="python3 tools/gemm_bench.py"
run() { timeout -k 10 200 $B > $O/$1.log 2>&1 && IMPLS=2 ROUNDS=2 timeout -k 10 200 $G > $O/$1_gemm.log 2>&1; }
rebuild() { touch simple-multimodal_amd/csrc/gemm2.hip && make -C simple-multimodal_amd/csrc EXTRA="$1" > $O/build_$2.log 2>&1; }
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad or lazy" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
run pipe_1 && rebuild -DMMF_TN_PIPE=0 old && run old_1 && rebuild -DMMF_TN_PIPE=1 new && run pipe_2 && rebuild -DMMF_TN_PIPE=0 old2 && run old_2 && rebuild -DMMF_TN_PIPE=1 new2 || exit 1
for f in $O/pipe_?.log $O/old_?.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"achieved": [0-9.]*' $f | head -1); done
grep -h "TN" $O/*_gemm.log
