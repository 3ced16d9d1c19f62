# HIP API time per identify_speaker_list call on a short clip (run from the repository root on the GPU box)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/idtrace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ONLY=${ONLY:-1} timeout -s KILL 120 rocprofv3 --hip-trace --stats -d $O/t --output-format csv -- python3 $R/tools/identify_latency.py > $O/run.log 2>&1
f=$(find $O/t -name "*hip_api_stats.csv" | head -1)
head -25 $f
