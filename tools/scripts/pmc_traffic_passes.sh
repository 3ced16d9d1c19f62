set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_f32 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/fetch_f32.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_f32 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/write_f32.log 2>&1
export STREAMZ_GEMM=bf16x3
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_x3 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/fetch_x3.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_x3 --output-format csv -- $R/tools/net_bench 6614000 16 1 > $O/write_x3.log 2>&1
unset STREAMZ_GEMM
find $O -name "*counter_collection.csv" | head
