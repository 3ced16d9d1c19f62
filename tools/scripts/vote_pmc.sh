# matrix-pipe busy fraction and wait counters of the vote kernel at one large batch, both wave layouts
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/votepmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ROWS=6614000
for W in 8 4; do
  export STREAMZ_VOTE_WAVES=$W
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $O/w$W --output-format csv -- python3 $R/tools/vote_timing.py > $O/w$W.log 2>&1
  python3 $R/tools/pmc_kernel.py $(find $O/w$W -name "*counter_collection.csv" | head -1) infer_vote_kernel > $O/w$W.txt
done
cat $O/w8.txt $O/w4.txt
