set -e
cd $GRAFT_REPO_ROOT
for c in 1572864 3307008 6614016; do
  echo "chunk $c" >> gpurun_out/chunk_try.txt
  STREAMZ_CHUNK_ROWS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 3 >> gpurun_out/chunk_try.txt 2>&1
done
