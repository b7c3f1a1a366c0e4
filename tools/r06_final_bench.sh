#!/bin/bash
# Round 6's committed bench evidence (on the GPU box): the driver's command, the same command under rocprofv3 --kernel-trace (summary
# through tools/rocpd_stats.py, with the timed region's own table), the Zipf and configs[4]-shape lines.  Outputs under gpurun_out/r06z/.
cd /root/repo
out=gpurun_out/r06z; mkdir -p $out
export TMPDIR=/tmp
# a fresh box measures the HBM-bound score kernel ~10 % slow for its first minute (NOTES round 5): four short runs first
for i in 1 2 3 4; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mrr --batch-sweep "" --traffic off --standalone-steps 0 --cold-items 0 > /dev/null 2>&1
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 2>$out/bench_err.log | tail -n 1 > $out/bench_line_driver_command.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/r06prof -o run -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-mrr --traffic off --batch-sweep "" --cold-items 0 --standalone-steps 0 > /tmp/r06prof.log 2>&1 )
db=$(find /tmp/r06prof -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" $out/default_bench_kernel_stats.md --window score_kernel 3 20 > /dev/null; python tools/rocpd_timeline.py "$db" > $out/step_timeline_b8192.txt 2>/dev/null; fi
grep '^{"metric"' /tmp/r06prof.log | tail -n 1 > $out/bench_line_under_rocprof.json
timeout 600 python bench.py --steps 20 --warmup 3 --item-distribution zipf --no-cpu-baseline 2>>$out/bench_err.log | tail -n 1 > $out/bench_line_zipf.json
timeout 900 python bench.py --steps 10 --warmup 2 --model ewma --loss hinge --dim 256 --items 10000000 --no-cpu-baseline --no-mrr --cold-items 0 --batch-sweep "" 2>>$out/bench_err.log | tail -n 1 > $out/bench_line_ewma256_10M_items.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r06prof_e -o run -- python /root/repo/bench.py --steps 10 --warmup 2 --model ewma --loss hinge --dim 256 --items 10000000 --no-cpu-baseline --no-mrr --traffic off --batch-sweep "" --cold-items 0 --standalone-steps 0 > /tmp/r06prof_e.log 2>&1 )
db=$(find /tmp/r06prof_e -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" $out/ewma256_kernel_stats.md --window ewma_seq_kernel 2 10 > /dev/null; fi
python - <<'PY'
import json
for f in ("bench_line_driver_command","bench_line_zipf","bench_line_ewma256_10M_items"):
    try:
        j=json.load(open(f"gpurun_out/r06z/{f}.json")); r=j.get("roofline") or {}
        print(f, round(j["value"]/1e6,2),"M/s", round(j["ms_per_step"],3),"ms; roofline frac", r.get("frac"), "traffic", r.get("traffic"), "cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("mode"))
    except Exception as e: print(f, "ERR", e)
PY
tail -40 $out/default_bench_kernel_stats.md
