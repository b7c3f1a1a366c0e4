# A/B of the streaming-gather build variants (tools/build_variant.sh): tools/r05_nt_ab.sh "<bench args>" variant...
cd /root/repo
mkdir -p gpurun_out/r05n
ARGS=$1; shift
Q="--no-cpu-baseline --no-mrr --batch-sweep= --traffic off --standalone-steps 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 $Q --cold-items 0 > /dev/null 2>&1; done
for rep in 1 2 3; do
for v in "$@"; do
  lib=/root/repo/sbr_rs_amd/libsbr_hip_$v.so; [ "$v" = base ] && lib=/root/repo/sbr_rs_amd/libsbr_hip.so
  SBR_SCORE_FORM=lockstep SBR_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 $Q $ARGS 2>gpurun_out/r05n/err_$v.log | tail -n 1 > gpurun_out/r05n/line_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05n/line_$v.json"))
r=d["roofline"]; c=d.get("roofline_cold") or {}
k={n: round(v["ms_per_launch"],3) for n,v in d.get("kernels",{}).items()}
print("$v", round(d["value"]/1e6,2), "M/s", round(d["ms_per_step"],4), "ms; score", round(r["avg_launch_ms"]*1e3,1), "us frac", round(r["frac"],3), "cold", round(c.get("frac",0),3), k)
PY
done; done
