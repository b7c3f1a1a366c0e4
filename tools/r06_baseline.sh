#!/bin/bash
# round 6: the state of the tree at the start of the second session: full GPU suite, the partitioned group driver line, the headline line
set -x
mkdir -p gpurun_out/r06
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/tests_all.log 2>&1
tail -5 gpurun_out/r06/tests_all.log
timeout 900 python bench.py --driver group --gpus 8 --partition-table --model ewma --loss hinge --dim 256 --items 10000000 --batch-sequences 8192 --steps 20 --warmup 3 > gpurun_out/r06/group8_partitioned.log 2>&1
tail -1 gpurun_out/r06/group8_partitioned.log > gpurun_out/r06/group_driver_partitioned_configs4.jsonl
tail -1 gpurun_out/r06/group8_partitioned.log | cut -c1-900
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r06/bench_default.log 2>&1
tail -1 gpurun_out/r06/bench_default.log > gpurun_out/r06/bench_line_driver_command.json
tail -1 gpurun_out/r06/bench_default.log | cut -c1-1500
