#!/usr/bin/env python
"""Per-dispatch values of one counter for every kernel whose name contains a substring, from a rocprofv3 rocpd database
(`--kernel-trace --pmc X`): prints JSON {kernel substring: [values in dispatch order]} (counter summed over instances).

    tools/pmc_dispatches.py run_results.db FETCH_SIZE score_kernel cal_read_seq_16B ...
"""
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


if __name__ == "__main__":
    db, counter = sys.argv[1], sys.argv[2]
    print(json.dumps({k: per_dispatch(db, counter, k) for k in sys.argv[3:]}))
