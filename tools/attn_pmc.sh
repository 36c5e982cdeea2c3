# SQ counters of the attention kernels alone at 1 / 2 / 3 workgroups per CU (tools/attn_bench.py, plain launches).
#   gpurun -- 'bash tools/attn_pmc.sh fwd' ; python tools/attn_pmc_summary.py gpurun_out/attn_pmc
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
WHAT=${1:-fwd}
O=gpurun_out/attn_pmc
rm -rf $O && mkdir -p $O
export MMF_ATTN_IMPLS=2 MMF_ATTN_CASES="t<-a" MMF_ATTN_NOGRAPH=1
for b in 8 16 24; do
  export MMF_ATTN_B=$b
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $O/b${b}_s1 --output-format csv -- python3 tools/attn_bench.py $WHAT > $O/b${b}_s1.log 2>&1 &&
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d $O/b${b}_s2 --output-format csv -- python3 tools/attn_bench.py $WHAT > $O/b${b}_s2.log 2>&1 &&
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/b${b}_s3 --output-format csv -- python3 tools/attn_bench.py $WHAT > $O/b${b}_s3.log 2>&1 || echo "pass failed b=$b"
done
