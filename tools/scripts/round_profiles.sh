# Everything profiles/r02_* is made from, in one pass on the GPU box (run from the repository root).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --gemm bf16x3 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --hidden1 2048 --hidden2 2048 --speakers 256 --gemm bf16x3 > $O/bench_wide_bf16x3.json 2> $O/bench_wide.err
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --hidden1 2048 --hidden2 2048 --speakers 256 > $O/bench_wide_f32.json 2>> $O/bench_wide.err
timeout -k 10 120 python tools/host_call_timing.py > $O/host_call_timing.txt 2>&1
STREAMZ_STEP_PROF=1 timeout -k 10 120 python tools/batch8_timing.py > $O/batch8_timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/stats_f32 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_f32.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/stats_x3 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --gemm bf16x3 > $O/stats_x3.log 2>&1
find $O -name "*kernel_stats.csv"
