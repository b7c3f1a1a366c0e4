cd /root/repo
mkdir -p gpurun_out/r06
timeout 900 python bench.py --driver group --gpus 8 --partition-table --model ewma --loss hinge --dim 256 --items 10000000 --batch-sequences 8192 --steps 20 --warmup 3 > gpurun_out/r06/group8_partitioned.log 2>&1
tail -1 gpurun_out/r06/group8_partitioned.log > gpurun_out/r06/group_driver_partitioned_configs4.jsonl
tail -1 gpurun_out/r06/group8_partitioned.log | cut -c1-600
bash tools/r06_ewma_ab.sh
