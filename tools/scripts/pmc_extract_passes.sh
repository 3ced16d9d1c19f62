# Counter passes of the extraction kernels (run from the repository root on the GPU box):
# HBM bytes (FETCH_SIZE, WRITE_SIZE: separate runs) and one pass of eight SQ counters.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02x
mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/write.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $O/sq --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/sq.log 2>&1
cd $R && python tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt
cat $O/summary.txt
