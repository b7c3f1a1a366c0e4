#!/bin/bash
# round 6: the whole-sequence EWMA form on two workgroups per CU adopted: the GPU suite, the configs[4]-shaped line with the PMC
# traffic of every kernel of the step, the d = 128 line, the partitioned group driver
mkdir -p gpurun_out/r06
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r06/tests_all_b.log 2>&1
tail -3 gpurun_out/r06/tests_all_b.log
timeout 900 python bench.py --model ewma --loss hinge --dim 256 --items 10000000 --steps 10 --warmup 2 --no-cpu-baseline --no-mrr --batch-sweep= --cold-items 0 > gpurun_out/r06/ewma256_line_b.log 2>&1
tail -1 gpurun_out/r06/ewma256_line_b.log > gpurun_out/r06/bench_line_ewma256_10M_items_whole.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_line_ewma256_10M_items_whole.json"))
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("step_bytes"), indent=1)[:3000])
print(json.dumps({k: v for k, v in (d.get("roofline") or {}).items() if k != "traffic_source"}, indent=1)[:3000])
print(json.dumps(d.get("kernels"))[:1000])
PY
timeout 600 python bench.py --model ewma --loss hinge --dim 128 --items 1000000 --steps 10 --warmup 2 --no-cpu-baseline --no-mrr --batch-sweep= --cold-items 0 --traffic off 2>&1 | tail -1 > gpurun_out/r06/bench_line_ewma128_1M_items_whole.json
cut -c1-300 gpurun_out/r06/bench_line_ewma128_1M_items_whole.json
timeout 900 python bench.py --driver group --gpus 8 --partition-table --model ewma --loss hinge --dim 256 --items 10000000 --batch-sequences 8192 --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/r06/group_driver_partitioned_configs4_whole.jsonl
cut -c1-400 gpurun_out/r06/group_driver_partitioned_configs4_whole.jsonl
