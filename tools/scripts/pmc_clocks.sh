# Clock and matrix-pipe busy fraction per training kernel (run from the repository root on the GPU box); the
# environment of the caller reaches net_bench, e.g. STREAMZ_NO_K64=1 bash tools/scripts/pmc_clocks.sh out_name
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/clocks_${1:-run}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/t --output-format csv -- $R/tools/net_bench 6614000 16 4 > $O/run.log 2>&1
cd $R
python3 tools/pmc_mfma_busy.py $(find $O/t -name "*counter_collection.csv" | head -1) $O/busy.json
rm -rf $O/t
