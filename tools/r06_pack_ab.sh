#!/bin/bash
# round 6: is the configs[4]-shaped EWMA line (two 5.7 ms steps per epoch) waiting for the host's epoch packing?  4 against 16 packing threads
mkdir -p gpurun_out/r06
cd /root/repo
run() {
  line=$(env $3 python bench.py --steps 10 --warmup 2 --model ewma --loss hinge --dim 256 --items 10000000 --no-cpu-baseline --no-mrr --standalone-steps 0 --cold-items 0 --batch-sweep '' --traffic off $2 2>/dev/null | tail -1)
  python - "$1" "$line" <<'PY' | tee -a gpurun_out/r06/pack_ab.jsonl
import json, sys
d = json.loads(sys.argv[2])
k = d["kernels"]
print(json.dumps({"run": sys.argv[1], "ms_per_step": round(d["ms_per_step"], 3), "M_per_s": round(d["value"] / 1e6, 1), "kernels_sum": round(k["SCORE"]["ms_per_launch"] + k["SPARSE_UPDATE"]["ms_per_launch"] + k.get("RECURRENT_BWD", {"ms_per_launch": 0})["ms_per_launch"], 3), "epoch_prepare_ms": round(d["epoch_prepare_ms"], 2)}))
PY
}
for i in 1 2 3 4 5; do run "4 packing threads"; run "16 packing threads" "" "SBR_HIP_LIB=$PWD/sbr_rs_amd/libsbr_hip_pack16.so"; done
