cd /root/repo
mkdir -p gpurun_out/r05d
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "one_sequence_steps or step_runs or movielens_fit or batch_of_one or ragged or popular or save_load or committed_vectors" ) > gpurun_out/r05d/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05d/pytest.log
( timeout 600 python tools/criterion_bench.py --oracle --batches 1 ) > gpurun_out/r05d/criterion.log 2>&1
( timeout 600 python tools/time_small_steps.py movielens 10 128 ) > gpurun_out/r05d/movielens.log 2>&1
( python tools/phase_clocks.py criterion; python tools/phase_clocks.py movielens ) > gpurun_out/r05d/phases.log 2>&1
tail -n 6 gpurun_out/r05d/pytest.log; cat gpurun_out/r05d/criterion.log | cut -c1-300; cat gpurun_out/r05d/movielens.log gpurun_out/r05d/phases.log
