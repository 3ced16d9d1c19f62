# HIP API time of a python tool (run from the repository root on the GPU box): bash tools/scripts/hip_api_stats.sh tools/<tool>.py
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hipapi
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --hip-trace --stats -d $O/t --output-format csv -- python3 $R/$1 > $O/run.log 2>&1
f=$(find $O/t -name "*hip_api_stats.csv" | head -1)
head -${2:-14} $f | cut -c1-140
tail -n 3 $O/run.log
