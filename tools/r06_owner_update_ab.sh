#!/bin/bash
# round 6: owner_update_kernel and owner_list_apply_kernel with every contribution of a row requested together (two dependent round trips
# per row instead of up to ten / 2 ndev + 2)
mkdir -p gpurun_out/r06
cd /root/repo
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_distributed_gpu.py -x -q -m gpu -k "multi_device or group or adam_multi or two_devices or distributed or rccl or peer or committed or save_load or num_threads or simulated or ranks or partition or api" > gpurun_out/r06/tests_owner_update.log 2>&1
tail -3 gpurun_out/r06/tests_owner_update.log
for rep in 1 2; do
  for lib in "" "$PWD/sbr_rs_amd/libsbr_hip_ou_before.so"; do
    SBR_HIP_LIB=$lib timeout 600 python bench.py --simulate-world 8 --steps 10 --warmup 2 2>/dev/null | tail -1 > /tmp/line.json
    python - "$lib" <<'PY' | tee -a gpurun_out/r06/owner_update_ab.jsonl
import json, sys
d = json.load(open("/tmp/line.json"))
print(json.dumps({"lib": sys.argv[1].split("/")[-1] or "shipped", "ms_per_phase": {k: round(v, 3) for k, v in d["ms_per_phase"].items()}, "exchange_kernels_ms": round(d["exchange_kernels_ms"], 3)}))
PY
  done
done
for lib in "" "$PWD/sbr_rs_amd/libsbr_hip_ou_before.so"; do
  SBR_HIP_LIB=$lib timeout 600 python bench.py --driver group --gpus 8 --steps 20 --warmup 3 2>/dev/null | tail -1 > /tmp/line.json
  python - "$lib" <<'PY' | tee -a gpurun_out/r06/owner_update_ab.jsonl
import json, sys
d = json.load(open("/tmp/line.json"))
print(json.dumps({"lib": sys.argv[1].split("/")[-1] or "shipped", "group_driver_ms_per_replica_step": round(d["ms_per_step"] / 8, 3), "M_per_s": round(d["value"] / 1e6, 1)}))
PY
done
for rep in 1 2; do
for lib in "" "$PWD/sbr_rs_amd/libsbr_hip_ou_before.so"; do
  SBR_HIP_LIB=$lib timeout 900 python bench.py --driver group --gpus 8 --partition-table --model ewma --loss hinge --dim 256 --items 10000000 --batch-sequences 8192 --steps 20 --warmup 3 2>/dev/null | tail -1 > /tmp/line.json
  python - "$lib" <<'PY' | tee -a gpurun_out/r06/owner_update_ab.jsonl
import json, sys
d = json.load(open("/tmp/line.json"))
print(json.dumps({"lib": sys.argv[1].split("/")[-1] or "shipped", "partitioned_group_driver_ms_per_step_all_replicas": round(d["ms_per_step"], 3), "M_per_s": round(d["value"] / 1e6, 1)}))
PY
done
done
