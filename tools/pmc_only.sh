cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01p
mkdir -p $O
export MMF_MULT_STREAMS=1 MMF_HIER_STREAMS=0
PM="python3 bench.py --steps 2 --warmup 1 --no-graph --profile-steps 1 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $PM > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $PM > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
