#!/bin/bash
# HBM traffic of the timed dispatches of one kernel at the bench's operating point, from two rocprofv3 PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs, --kernel-trace only beside them; MI355X_MICROARCH.md "HBM"), appended as one entry to a JSON
# list file.  On the GPU box:
#   tools/score_traffic_profile.sh OUT.json KERNEL_SUBSTRING WARMUP_DISPATCHES LABEL [bench args...]
# e.g. tools/score_traffic_profile.sh gpurun_out/traffic.json score_kernel 5 warm --steps 20 --warmup 5
out=$1; kernel=$2; skip=$3; label=$4; shift 4
export TMPDIR=/tmp
tmp=$(mktemp -d /tmp/pmcXXXX)
args="--no-cpu-baseline --no-mrr --batch-sweep= --standalone-steps 0 --cold-items 0 $*"
( cd /tmp && true )
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $tmp/f -o run -- python bench.py $args > $tmp/f.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $tmp/w -o run -- python bench.py $args > $tmp/w.log 2>&1
fdb=$(find $tmp/f -name '*_results.db' | head -1); wdb=$(find $tmp/w -name '*_results.db' | head -1)
python - "$fdb" "$wdb" "$kernel" "$skip" "$label" "$out" "$tmp/f.log" "$args" <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from pmc_dispatches import per_dispatch
fdb, wdb, kernel, skip, label, out, log, args = sys.argv[1:9]
skip = int(skip)
f = per_dispatch(fdb, "FETCH_SIZE", kernel)[skip:]
w = per_dispatch(wdb, "WRITE_SIZE", kernel)[skip:]
n = min(len(f), len(w))
line = json.loads([x for x in open(log).read().splitlines() if x.startswith('{"metric"')][-1])  # (rocprofv3 prints after the app)
roof = line.get("roofline") or {}
rows, k, d = roof.get("rows_per_launch"), roof.get("mean_negatives_scored"), line["config"]["dim"]
entry = {"label": label, "kernel": kernel, "bench_args": args, "dispatches_averaged": n, "warmup_dispatches_skipped": skip,
         "rows_per_launch": rows, "mean_negatives_scored": k, "items": line["config"]["items"], "dim": d,
         "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
         "FETCH_SIZE_KiB_mean": sum(f[:n]) / max(n, 1), "WRITE_SIZE_KiB_mean": sum(w[:n]) / max(n, 1)}
up = (2.0 * entry["FETCH_SIZE_KiB_mean"] + entry["WRITE_SIZE_KiB_mean"]) * 1024.0
entry["hbm_bytes_per_launch"] = up
if rows and k is not None:  # isolated 4-byte bias reads may be counted at a full 64-B sector each, i.e. not halved: lower figure
    entry["hbm_bytes_per_launch_lower"] = up - 64.0 * (1 + k) * rows
    if entry["algorithmic_bytes_per_launch"]:
        entry["traffic_over_algorithmic"] = [entry["hbm_bytes_per_launch_lower"] / entry["algorithmic_bytes_per_launch"], up / entry["algorithmic_bytes_per_launch"]]
entry["ms_per_step_profiled"] = line.get("ms_per_step")
entry["kernel_avg_launch_ms_profiled"] = roof.get("avg_launch_ms")
cur = json.load(open(out)) if os.path.exists(out) else []
cur.append(entry)
json.dump(cur, open(out, "w"), indent=1)
print(json.dumps(entry))
PY
rm -rf $tmp
