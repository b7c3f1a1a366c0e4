cd /root/repo
mkdir -p gpurun_out/r05g
( timeout 2400 python -m pytest tests -x -q -m gpu --durations=10 ) > gpurun_out/r05g/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05g/pytest_gpu.log
tail -n 4 gpurun_out/r05g/pytest_gpu.log
