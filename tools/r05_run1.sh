cd /root/repo
mkdir -p gpurun_out/r05a
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "group or partitioned" --durations=15 ) > gpurun_out/r05a/pytest_group.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05a/pytest_group.log
( timeout 600 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k "group_driver" ) > gpurun_out/r05a/pytest_dist.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05a/pytest_dist.log
for n in 1 2 4 8; do
  timeout 600 python bench.py --driver group --gpus $n --steps 20 --warmup 3 2>>gpurun_out/r05a/group_err.log | tail -1 >> gpurun_out/r05a/group_driver.jsonl
done
timeout 900 python bench.py --driver group --gpus 8 --partition-table --model ewma --loss hinge --dim 256 --items 10000000 --max-len 64 --batch-sequences 8192 --steps 10 --warmup 2 2>>gpurun_out/r05a/group_err.log | tail -1 >> gpurun_out/r05a/group_partitioned.jsonl
tail -3 gpurun_out/r05a/pytest_group.log gpurun_out/r05a/pytest_dist.log
