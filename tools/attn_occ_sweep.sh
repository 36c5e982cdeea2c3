cd $GRAFT_REPO_ROOT
export MMF_ATTN_IMPLS=2 MMF_ATTN_CASES="t<-a,t<-t"
for b in 2 4 8 12 16 24 32 48 64; do echo "B=$b"; MMF_ATTN_B=$b python tools/attn_bench.py both; done
