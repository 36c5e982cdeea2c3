cd $GRAFT_REPO_ROOT
export MMF_ATTN_IMPLS=2 MMF_ATTN_CASES="cross,self,t<-a,t<-t,v<-t,grpA"
for r in 1 2; do
echo "old"; MMF_LIB_PATH=$PWD/simple-multimodal_amd/mmfusion/libmmfusion_old.so timeout -k 10 100 python tools/attn_bench.py fwd 2>&1 | grep -E "^(fwd|bwd)"
echo "new"; timeout -k 10 100 python tools/attn_bench.py fwd 2>&1 | grep -E "^(fwd|bwd)"
done
