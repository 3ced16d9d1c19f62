# One pass of SQ counters over the training kernels (run from the repository root on the GPU box):
# matrix-pipe busy cycles, wave cycles, waiting cycles, instruction counts per kernel.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02s
mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $O/sq_f32 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/sq_f32.log 2>&1
cd $R && python tools/pmc_summary.py $(find $O/sq_f32 -name "*counter_collection.csv") > $O/summary_f32.txt
cat $O/summary_f32.txt
