cd /root/repo
mkdir -p gpurun_out/r05b
( timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 ) > gpurun_out/r05b/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05b/pytest_gpu.log
for n in 1 2 4 8; do
  timeout 600 python bench.py --driver group --gpus $n --steps 20 --warmup 3 2>>gpurun_out/r05b/group_err.log | tail -n 1 >> gpurun_out/r05b/group_driver.jsonl
done
tail -n 5 gpurun_out/r05b/pytest_gpu.log
