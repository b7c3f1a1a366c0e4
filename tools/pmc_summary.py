#!/usr/bin/env python
"""Per-kernel PMC summary from the rocpd databases tools/pmc_collect.sh leaves (one per counter group):

    tools/pmc_summary.py gpurun_out/r02_pmc_base [kernel substring ...] > profiles/rNN_pmc.md

For every kernel whose name contains one of the substrings: mean over its dispatches of every counter
(summed over the hardware instances rocprofv3 reports) and the mean dispatch duration."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    pats = sys.argv[2:] or ["lstm_fwd_seq", "lstm_bwd_seq", "lstm_dw", "score_kernel", "seg_short"]
    table = defaultdict(dict)  # kernel -> counter -> mean
    dur = {}
    for dbp in sorted(glob.glob(os.path.join(root, "*", "run_results.db"))):
        db = sqlite3.connect(dbp)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        for pat in pats:
            rows = db.execute(f"select counter_name, dispatch_id, sum(value) from counters_collection where {name_col} like ? "
                              "group by counter_name, dispatch_id", (f"%{pat}%",)).fetchall()
            per = defaultdict(list)
            for cn, _, v in rows:
                per[cn].append(v)
            for cn, vs in per.items():
                table[pat][cn] = (sum(vs) / len(vs), len(vs))
            try:
                d = db.execute("select avg(end - start), count(*) from kernels where name like ?", (f"%{pat}%",)).fetchone()
                if d and d[0]:
                    dur[pat] = (d[0] / 1e3, d[1])
            except sqlite3.Error:
                pass
    print("| kernel | counter | mean per dispatch | dispatches |")
    print("|---|---|---|---|")
    for pat in pats:
        if pat in dur:
            print(f"| {pat} | duration_us (under PMC collection) | {dur[pat][0]:.1f} | {dur[pat][1]} |")
        for cn in sorted(table[pat]):
            v, n = table[pat][cn]
            print(f"| {pat} | {cn} | {v:.4g} | {n} |")


if __name__ == "__main__":
    main()
