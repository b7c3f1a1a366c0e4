cd /root/repo
mkdir -p gpurun_out/r05s
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "warp or whole_fit or intermediates or bench_regime" ) > gpurun_out/r05s/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05s/pytest.log
tail -n 5 gpurun_out/r05s/pytest.log
Q="--no-cpu-baseline --no-mrr --batch-sweep= --traffic off --standalone-steps 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 $Q > /dev/null 2>&1; done
for form in lockstep refill lockstep refill; do
  SBR_SCORE_FORM=$form python bench.py --steps 20 --warmup 5 $Q 2>gpurun_out/r05s/err_$form.log | tail -n 1 > gpurun_out/r05s/line_$form.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05s/line_$form.json"))
print("$form", d["value"], d["ms_per_step"], json.dumps(d["roofline"]), json.dumps(d.get("roofline_cold")))
PY
done
