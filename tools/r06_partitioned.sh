#!/bin/bash
# round 6: the partitioned-table step without the host in the loop
set -x
mkdir -p gpurun_out/r06
cd /root/repo
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_distributed_gpu.py -x -q -m gpu -k "partition or group_plan or api" > gpurun_out/r06/tests_partitioned.log 2>&1
tail -5 gpurun_out/r06/tests_partitioned.log
timeout 900 python bench.py --driver group --gpus 8 --partition-table --model ewma --loss hinge --dim 256 --items 10000000 --batch-sequences 8192 --steps 20 --warmup 3 > gpurun_out/r06/group8_partitioned.log 2>&1
tail -1 gpurun_out/r06/group8_partitioned.log > gpurun_out/r06/group_driver_partitioned_configs4.jsonl
tail -1 gpurun_out/r06/group8_partitioned.log | cut -c1-900
