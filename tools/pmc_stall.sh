# Wave-stall / LDS / L2 counters per kernel of the eager MulT step (one counter set per pass).
#   gpurun -- 'bash tools/pmc_stall.sh && python3 tools/pmc_stall_summary.py gpurun_out/r01s profiles/<round>_pmc_stalls.txt'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r01}s
rm -rf $O && mkdir -p $O
export MMF_MULT_STREAMS=1 MMF_HIER_STREAMS=0
rocprofv3 -L > $O/avail.txt 2>&1
PM="python3 bench.py --steps 2 --warmup 1 --no-graph --profile-steps 1 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $O/sq1 --output-format csv -- $PM > $O/sq1.log 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/tcc --output-format csv -- $PM > $O/tcc.log 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $O/sq2 --output-format csv -- $PM > $O/sq2.log 2>&1
