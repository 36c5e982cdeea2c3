cd $GRAFT_REPO_ROOT
export MMF_ATTN_IMPLS=4 MMF_ATTN_CASES="t<-a"
for d in 7 15 23 39 63; do echo "debug=$d"; MMF_ATTN2_DEBUG=$d timeout -k 10 60 python tools/attn_bench.py fwd 2>&1 | grep fwd; done
