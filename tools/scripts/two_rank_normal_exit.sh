set -x
mkdir -p gpurun_out/tworank
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2
RANK=0 LOCAL_RANK=0 timeout -k 10 200 python -X faulthandler tests/two_rank_child.py gpurun_out/tworank > gpurun_out/tworank/r0.log 2>&1 &
P0=$!
RANK=1 LOCAL_RANK=1 timeout -k 10 200 python -X faulthandler tests/two_rank_child.py gpurun_out/tworank > gpurun_out/tworank/r1.log 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?" | tee -a gpurun_out/tworank/rc.txt
wait $P1; echo "rank1 rc=$?" | tee -a gpurun_out/tworank/rc.txt
tail -40 gpurun_out/tworank/r0.log gpurun_out/tworank/r1.log
