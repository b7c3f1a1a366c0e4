#!/usr/bin/env python
"""HBM traffic of one kernel from two rocprofv3 PMC passes (rocpd databases):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d A -o run -- python bench.py --no-cpu-baseline --no-mrr
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d B -o run -- python bench.py --no-cpu-baseline --no-mrr
    tools/pmc_traffic.py A/run_results.db B/run_results.db score_kernel 3 > profiles/r02_score_kernel_pmc.json   (round 2; round 3: tools/pmc_dispatches.py + profiles/score_kernel_traffic.json)

Counters are per dispatch; argument 4 = number of leading (warm-up) dispatches to skip.  Corrections
per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of 16-byte-per-lane loads, which is how this kernel reads every table
and hidden-state row, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import json
import sqlite3
import sys


def per_dispatch(db_path, counter, kernel):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(f"select dispatch_id, sum(value) from counters_collection where counter_name = ? and {name_col} like ? "
                      "group by dispatch_id order by dispatch_id", (counter, f"%{kernel}%")).fetchall()
    return [v for _, v in rows]


def main():
    fetch_db, write_db, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    f = per_dispatch(fetch_db, "FETCH_SIZE", kernel)[skip:]
    w = per_dispatch(write_db, "WRITE_SIZE", kernel)[skip:]
    n = min(len(f), len(w))
    bytes_per = [(2.0 * f[i] + w[i]) * 1024.0 for i in range(n)]
    print(json.dumps({
        "kernel": kernel, "dispatches_averaged": n, "warmup_dispatches_skipped": skip,
        "FETCH_SIZE_KiB_mean": sum(f[:n]) / max(n, 1), "WRITE_SIZE_KiB_mean": sum(w[:n]) / max(n, 1),
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB units; gfx950 FETCH_SIZE counts half of 16-B-per-lane loads)",
        "hbm_bytes_per_launch": sum(bytes_per) / max(n, 1),
    }, indent=1))


if __name__ == "__main__":
    main()
