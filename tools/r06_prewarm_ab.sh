#!/bin/bash
# round 6: does the bench's short timed region (20 steps of ~2.4 ms) see a GPU that is still waking up?  The same command with and
# without a model-neutral busy period before the warm-up steps, alternating, on one box.
mkdir -p gpurun_out/r06
cd /root/repo
out=gpurun_out/r06/prewarm_ab.jsonl
run() { # label, extra args
  line=$(python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-mrr --standalone-steps 0 --cold-items 0 --batch-sweep '' --traffic off $2 2>/dev/null | tail -1)
  python - "$1" "$line" <<'PY' | tee -a gpurun_out/r06/prewarm_ab.jsonl
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"run": sys.argv[1], "ms_per_step": round(d["ms_per_step"], 4), "M_per_s": round(d["value"] / 1e6, 2), "score_us": round(1e3 * d["roofline"]["avg_launch_ms"], 2),
                  "frac": round(d["roofline"]["frac"], 4), "k": round(d["roofline"]["mean_negatives_scored"], 3)}))
PY
}
for rep in 1 2 3; do
  run "lstm prewarm 0" "--prewarm-seconds 0"
  run "lstm prewarm 2" "--prewarm-seconds 2"
  sleep 20
  run "lstm prewarm 0 after 20 s idle" "--prewarm-seconds 0"
  run "lstm prewarm 2 after nothing" "--prewarm-seconds 2"
done
STEPS=10
for rep in 1 2 3; do
  run "ewma prewarm 0" "--prewarm-seconds 0 --model ewma --loss hinge --dim 256 --items 10000000"
  run "ewma prewarm 2" "--prewarm-seconds 2 --model ewma --loss hinge --dim 256 --items 10000000"
done
