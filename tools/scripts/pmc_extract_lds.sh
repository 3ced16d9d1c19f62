# How busy is the LDS during extraction?  One counter pass of tools/extract_bench (run from the repository root on the GPU box).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/xlds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/a --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/b --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/b.log 2>&1
cd $R
python3 tools/pmc_kernel.py $(find $O/a -name "*counter_collection.csv" | head -1) mfcc_base_kernel > $O/lds.txt
python3 tools/pmc_kernel.py $(find $O/b -name "*counter_collection.csv" | head -1) mfcc_base_kernel >> $O/lds.txt
cat $O/lds.txt
