# memory-side counters of the vote kernel at one large batch
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/votepmc2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ROWS=6614000
export STREAMZ_VOTE_WAVES=8
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/a --output-format csv -- python3 $R/tools/vote_timing.py > $O/a.log 2>&1
python3 $R/tools/pmc_kernel.py $(find $O/a -name "*counter_collection.csv" | head -1) infer_vote_kernel > $O/a.txt
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $O/b --output-format csv -- python3 $R/tools/vote_timing.py > $O/b.log 2>&1
python3 $R/tools/pmc_kernel.py $(find $O/b -name "*counter_collection.csv" | head -1) infer_vote_kernel > $O/b.txt
cat $O/a.txt $O/b.txt
