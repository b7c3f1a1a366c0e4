#!/bin/bash
# A/B of one optimiser step under build variants / test hooks (DESIGN.md section 7 "Switches"; the round 1-4 A/B variables are gone): every line is one bench.py process,
#   tools/step_ab.sh OUT.jsonl "B1 B2 ..." "ENV1=a ENV2=b" "ENV1=c" ...
# prints batch, switches, ms per step and the per-family kernel times.  None of the switches changes a result bit.
out=$1; shift
batches=$1; shift
for envs in "" "$@"; do
  for b in $batches; do
    line=$(env $envs python bench.py --steps ${STEPS:-20} --warmup 5 --batch-sequences $b --no-cpu-baseline --no-mrr --standalone-steps 0 \
           --cold-items 0 --batch-sweep '' $EXTRA 2>/dev/null | tail -1)
    python - "$b" "$envs" "$line" <<'PY' | tee -a "$out"
import json, sys
b, envs, line = sys.argv[1:4]
try:
    d = json.loads(line)
    k = {n: round(v["ms_per_launch"], 3) for n, v in d["kernels"].items()}
    print(json.dumps({"batch": int(b), "env": envs, "ms_per_step": round(d["ms_per_step"], 4), "M_per_s": round(d["value"] / 1e6, 2), "kernels": k}))
except Exception as e:
    print(json.dumps({"batch": int(b), "env": envs, "error": repr(e), "line": line[-300:]}))
PY
  done
done
