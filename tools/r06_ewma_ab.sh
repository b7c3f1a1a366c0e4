#!/bin/bash
# round 6: the EWMA step at configs[4]'s shape (d = 256, 1e7 items, 50 000 sequences per step): the whole-sequence form (forward scan,
# scores and backward scan of a sequence by one wave) on all waves and on few enough waves that a sequence's rows may still be cached
# when its backward scan re-reads them; then the byte budget of the shipped step with the PMC traffic of every kernel.
mkdir -p gpurun_out/r06
cd /root/repo
out=gpurun_out/r06/ewma_whole_ab.jsonl
export EXTRA="--model ewma --loss hinge --dim 256 --items 10000000 --traffic off"
export STEPS=10
L=$PWD/sbr_rs_amd
for rep in 1 2; do
  bash tools/step_ab.sh $out "50000" "SBR_HIP_LIB=$L/libsbr_hip_whole.so" "SBR_HIP_LIB=$L/libsbr_hip_whole512.so" "SBR_HIP_LIB=$L/libsbr_hip_whole256.so" "SBR_HIP_LIB=$L/libsbr_hip_whole256u8.so" | cut -c1-400
done
timeout 900 python bench.py --model ewma --loss hinge --dim 256 --items 10000000 --steps 10 --warmup 2 --no-cpu-baseline --no-mrr --batch-sweep= --cold-items 0 > gpurun_out/r06/ewma256_line.log 2>&1
tail -1 gpurun_out/r06/ewma256_line.log > gpurun_out/r06/bench_line_ewma256_10M_items.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_line_ewma256_10M_items.json"))
print(json.dumps(d.get("step_bytes"), indent=1)[:3000])
print(json.dumps((d.get("roofline") or {}).get("traffic_step"), indent=1)[:2000])
PY
