cd /root/repo
mkdir -p gpurun_out/r05h
( timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "reference_order" --durations=10 ) > gpurun_out/r05h/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05h/pytest.log
tail -n 25 gpurun_out/r05h/pytest.log
