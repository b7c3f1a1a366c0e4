cd /root/repo
mkdir -p gpurun_out/r05s
Q="--no-cpu-baseline --no-mrr --batch-sweep= --traffic off --standalone-steps 0 --cold-items 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 $Q > /dev/null 2>&1; done
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 $Q $EXTRA 2>gpurun_out/r05s/err_$label.log | tail -n 1 > gpurun_out/r05s/line_$label.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05s/line_$label.json"))
r=d["roofline"]
print("$label", round(d["value"]/1e6,2), round(d["ms_per_step"],4), "score us", round(r["avg_launch_ms"]*1e3,1), "frac", round(r["frac"],3), "k", round(r["mean_negatives_scored"],3))
PY
}
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  run $label ${envs//,/ }
done
