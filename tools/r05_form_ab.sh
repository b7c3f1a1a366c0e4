# A/B of SBR_SCORE_FORM values on the shipping library in the step: tools/r05_form_ab.sh "<bench args>" form...
cd /root/repo
mkdir -p gpurun_out/r05n
ARGS=$1; shift
Q="--no-cpu-baseline --no-mrr --batch-sweep= --traffic off --standalone-steps 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 $Q --cold-items 0 > /dev/null 2>&1; done
for rep in 1 2 3; do
for v in "$@"; do
  SBR_SCORE_FORM=$v python bench.py --steps 20 --warmup 5 $Q $ARGS 2>gpurun_out/r05n/err_$v.log | tail -n 1 > gpurun_out/r05n/line_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05n/line_$v.json"))
r=d["roofline"]; c=d.get("roofline_cold") or {}
print("$v", round(d["value"]/1e6,2), "M/s", round(d["ms_per_step"],4), "ms; score", round(r["avg_launch_ms"]*1e3,1), "us frac", round(r["frac"],3), "cold", round(c.get("frac",0),3), "cold us", round(c.get("avg_launch_ms",0)*1e3,1))
PY
done; done
