cd /root/repo
mkdir -p gpurun_out/r05c
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "one_sequence_steps or step_runs or movielens or batch_of_one" --durations=8 ) > gpurun_out/r05c/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05c/pytest.log
( timeout 600 python tools/criterion_bench.py --oracle --batches 1 ) > gpurun_out/r05c/criterion.log 2>&1
( timeout 600 python tools/time_small_steps.py movielens 10 128 ) > gpurun_out/r05c/movielens.log 2>&1
tail -n 6 gpurun_out/r05c/pytest.log; cat gpurun_out/r05c/criterion.log | cut -c1-300; cat gpurun_out/r05c/movielens.log
