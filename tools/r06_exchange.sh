#!/bin/bash
# round 6: the owner-applied update of the replicated exchange against rounds 1-5's gradient all-gather (one GPU)
set -x
mkdir -p gpurun_out/r06
cd /root/repo
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_distributed_gpu.py -x -q -m gpu -k "multi_device or group or adam_multi or two_devices or distributed or rccl or peer or bench or committed or save_load or num_threads" > gpurun_out/r06/tests_exchange.log 2>&1
tail -5 gpurun_out/r06/tests_exchange.log
for ex in owner gradient; do
  timeout 600 python bench.py --driver group --gpus 8 --steps 20 --warmup 3 --exchange $ex > gpurun_out/r06/group8_$ex.log 2>&1
  tail -1 gpurun_out/r06/group8_$ex.log >> gpurun_out/r06/group_driver.jsonl
  timeout 600 python bench.py --simulate-world 8 --steps 10 --warmup 2 --exchange $ex > gpurun_out/r06/sim8_$ex.log 2>&1
  tail -1 gpurun_out/r06/sim8_$ex.log >> gpurun_out/r06/simulate_world8.jsonl
done
timeout 300 python bench.py --force-exchange --steps 10 --warmup 2 --no-cpu-baseline --no-mrr --batch-sweep= --standalone-steps 0 --cold-items 0 --traffic off > gpurun_out/r06/force_exchange.log 2>&1
tail -1 gpurun_out/r06/force_exchange.log | cut -c1-600
