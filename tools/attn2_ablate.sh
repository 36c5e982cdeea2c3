# MMF_ATTN2_DEBUG ablation of the default forward (1: no K/V DMA after tile 0, 2: waves stage tiles but do not compute)
cd $GRAFT_REPO_ROOT
export MMF_ATTN_IMPLS=2 MMF_ATTN_CASES="t<-a,t<-t,cross"
for d in 0 1 2 3; do echo "debug=$d"; MMF_ATTN2_DEBUG=$d timeout -k 10 60 python tools/attn_bench.py fwd 2>&1 | grep "^fwd"; done
