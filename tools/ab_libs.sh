#!/bin/bash
# Same-box A/B of library variants built by tools/build_variant.sh:  LIBS="base p1 d43" WHAT=both CASES="cross,self,t<-a" bash tools/ab_libs.sh
# For each variant: the attention kernel tests (parity first), then tools/attn_bench.py on the chosen cases, twice, interleaved.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-ab}; mkdir -p $O
export MMF_ATTN_IMPLS=2 MMF_ATTN_CASES="${CASES:-cross,self,t<-a,t<-t,v<-t,t<-v,narrow4,grpA}"
M=$PWD/simple-multimodal_amd/mmfusion
for v in $LIBS; do
  [ "$v" = base ] && continue
  if ! MMF_LIB_PATH=$M/libmmfusion_$v.so timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "${TESTK:-attention}" > $O/test_$v.log 2>&1; then
    echo "PARITY FAILED for $v"; tail -15 $O/test_$v.log; exit 1; fi
  tail -1 $O/test_$v.log
done
for r in 1 2; do for v in $LIBS; do
  echo "== $v (run $r)"; MMF_LIB_PATH=$M/libmmfusion_$v.so timeout -k 10 150 python tools/attn_bench.py ${WHAT:-bwd} 2>&1 | grep -E "^(fwd|bwd)"
done; done | tee $O/ab.log
