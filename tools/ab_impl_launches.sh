cd $GRAFT_REPO_ROOT
O=gpurun_out/abi; rm -rf $O; mkdir -p $O
export MMF_MULT_STREAMS=2
for cfg in auto impl2 impl4 impl5; do
  case $cfg in auto) E="";; impl2) E="MMF_GEMM_IMPL=2";; impl4) E="MMF_GEMM_IMPL=4";; impl5) E="MMF_GEMM_IMPL=5";; esac
  env $E timeout -k 10 200 python3 tools/step_launches.py mult > $O/$cfg.log 2>&1 || exit 1
done
for cfg in auto impl2 impl4 impl5; do echo "== $cfg"; grep "gemm" $O/$cfg.log | cut -c1-75; done
