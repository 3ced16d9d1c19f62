# Everything profiles/<tag>_* is made from, in one pass on the GPU box (run from the repository root):
#   bash tools/scripts/round_profiles.sh r03
# Every GPU step has its own time limit; the script stops at the first one that fails.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench done"
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --gemm bf16x3 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --hidden1 2048 --hidden2 2048 --speakers 256 --gemm bf16x3 > $O/bench_wide_bf16x3.json 2> $O/bench_wide.err
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --hidden1 2048 --hidden2 2048 --speakers 256 > $O/bench_wide_f32.json 2>> $O/bench_wide.err
echo "variants done"
timeout -k 10 120 python tools/vote_timing.py > $O/vote_timing.txt 2>&1
TILES=8192,65536,262144 timeout -k 10 200 python tools/lanes_timing.py > $O/identify_100h.txt 2>&1
echo "inference timings done"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/stats_f32 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_f32.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/stats_x3 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --gemm bf16x3 > $O/stats_x3.log 2>&1
echo "kernel stats done"
# HBM traffic of the training kernels: separate passes per counter
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_f32 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/fetch_f32.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_f32 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/write_f32.log 2>&1
# matrix-pipe busy fraction of the training kernels
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $O/sq_f32 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/sq_f32.log 2>&1
echo "training counters done"
# the extraction kernels
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/xfetch --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/xfetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/xwrite --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/xwrite.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $O/xsq --output-format csv -- $R/tools/extract_bench 1000 60 1 > $O/xsq.log 2>&1
timeout -k 10 60 $R/tools/extract_bench 1000 60 5 > $O/extract_bench.txt 2>&1
echo "extraction counters done"
# the vote kernel: matrix-pipe busy, HBM bytes
export ROWS=6614000
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/vote_sq --output-format csv -- python3 $R/tools/vote_timing.py > $O/vote_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/vote_fetch --output-format csv -- python3 $R/tools/vote_timing.py > $O/vote_fetch.log 2>&1
unset ROWS
cd $R
python3 tools/pmc_kernel.py $(find $O/vote_sq -name "*counter_collection.csv" | head -1) infer_vote_kernel > $O/vote_pmc.txt
python3 tools/pmc_kernel.py $(find $O/vote_fetch -name "*counter_collection.csv" | head -1) infer_vote_kernel >> $O/vote_pmc.txt
python3 tools/pmc_summary.py $(find $O/sq_f32 -name "*counter_collection.csv") > $O/pmc_train_sq.txt
python3 tools/pmc_summary.py $(find $O/xfetch $O/xwrite $O/xsq -name "*counter_collection.csv") > $O/pmc_extract_counters.txt
python3 tools/pmc_mfma_busy.py $(find $O/sq_f32 -name "*counter_collection.csv" | head -1) $O/pmc_mfma_busy.json > /dev/null
python3 tools/pmc_traffic.py $(find $O/fetch_f32 -name "*counter_collection.csv" | head -1) $(find $O/write_f32 -name "*counter_collection.csv" | head -1) 6614000 $O/f32_pmc_traffic.json > /dev/null
cp $(find $O/stats_f32 -name "*kernel_stats.csv" | head -1) $O/f32_kernel_stats.csv
cp $(find $O/stats_x3 -name "*kernel_stats.csv" | head -1) $O/bf16x3_kernel_stats.csv
cp $(find $O/fetch_f32 -name "*counter_collection.csv" | head -1) $O/f32_pmc_fetch_size.csv
cp $(find $O/write_f32 -name "*counter_collection.csv" | head -1) $O/f32_pmc_write_size.csv
# the raw traces are large; only the summaries travel back
rm -rf $O/stats_f32 $O/stats_x3 $O/fetch_f32 $O/write_f32 $O/sq_f32 $O/xfetch $O/xwrite $O/xsq $O/vote_sq $O/vote_fetch
ls -la $O
