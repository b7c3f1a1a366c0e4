#!/bin/bash
# round 6: run-to-run spread of the configs[4]-shaped EWMA line (HBM-bound, 20 GB of tables gathered at random): three processes at the
# start of a box's life, three after a series of LSTM bench processes, the shipped one-pass form and round 5's two launches alternating
mkdir -p gpurun_out/r06
cd /root/repo
out=gpurun_out/r06/ewma_spread.jsonl
run() {
  line=$(env $3 python bench.py --steps 10 --warmup 2 --model ewma --loss hinge --dim 256 --items 10000000 --no-cpu-baseline --no-mrr --standalone-steps 0 --cold-items 0 --batch-sweep '' --traffic off $2 2>/dev/null | tail -1)
  python - "$1" "$line" <<'PY' | tee -a gpurun_out/r06/ewma_spread.jsonl
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"run": sys.argv[1], "ms_per_step": round(d["ms_per_step"], 3), "M_per_s": round(d["value"] / 1e6, 1), "kernels": {n: round(v["ms_per_launch"], 3) for n, v in d["kernels"].items()}}))
PY
}
for i in 1 2 3; do run "start of box, one pass"; run "start of box, two launches (round 5 form)" "" "SBR_HIP_LIB=$PWD/sbr_rs_amd/libsbr_hip_r5form.so"; done
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mrr --batch-sweep "" --traffic off --standalone-steps 0 --cold-items 0 > /dev/null 2>&1; done
timeout 300 python bench.py --steps 20 --warmup 3 --item-distribution zipf --no-cpu-baseline --no-mrr --batch-sweep "" --traffic off --standalone-steps 0 --cold-items 0 > /dev/null 2>&1
for i in 1 2 3; do run "after seven LSTM processes, one pass"; run "after seven LSTM processes, two launches (round 5 form)" "" "SBR_HIP_LIB=$PWD/sbr_rs_amd/libsbr_hip_r5form.so"; done
