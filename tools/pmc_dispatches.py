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


def step_window(db_path, counter, kernel, first, count):
    """Every kernel's share of `counter` over `count` steps: the dispatches from the `first`-th dispatch of `kernel` (a substring; the
    kernel that marks a step) up to, not including, its (`first` + `count`)-th.  Returns {short kernel name: sum over the window}."""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(f"select dispatch_id, {name_col}, sum(value) from counters_collection where counter_name = ? "
                      f"group by dispatch_id, {name_col} order by dispatch_id", (counter,)).fetchall()
    marks = [i for i, (_, n, _) in enumerate(rows) if kernel in n]
    if len(marks) < first + count:
        return None
    out = {}
    for _, n, v in rows[marks[first]:marks[first + count] if len(marks) > first + count else len(rows)]:
        short = n.split("<")[0].split("(")[0].replace("void ", "").replace("sbr::", "").strip()
        out[short] = out.get(short, 0.0) + v
    return out


if __name__ == "__main__":
    db, counter = sys.argv[1], sys.argv[2]
    print(json.dumps({k: per_dispatch(db, counter, k) for k in sys.argv[3:]}))
