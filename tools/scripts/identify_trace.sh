# per-kernel durations inside the tiled identify (run from the repository root on the GPU box): TILES=8192 by default
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/idt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TILES=${TILES:-8192}
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python3 $R/tools/lanes_timing.py > $O/run.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
grep -E "Name|mfcc_base|finish_kernel|infer_vote" $f | cut -c1-220
