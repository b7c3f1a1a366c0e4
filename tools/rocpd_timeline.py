#!/usr/bin/env python
"""Timeline of one bench step from a rocprofv3 rocpd database (kernel trace): every dispatch between two
consecutive launches of the forward kernel, with start offsets, durations and the idle gaps of the critical path.
Usage: rocpd_timeline.py results.db [step-index]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
    fwd = [i for i, r in enumerate(rows) if "lstm_fwd_seq_kernel" in r[0] or "lstm_fwd_wave_kernel" in r[0]]
    a, b = fwd[which], fwd[which + 1]
    t0 = rows[a][1]
    print(f"step {which}: {(rows[b][1] - t0) / 1e6:.3f} ms from forward launch to forward launch; columns: start ms, duration ms, queue, kernel")
    for r in rows[a:b]:
        name = re.sub(r"\(.*$", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "").replace("sbr::", "")
        name = re.sub(r"rocprim::[A-Za-z_0-9:]*detail::", "rocprim::", name)[:70]
        print(f"{(r[1] - t0) / 1e6:9.3f} {(r[2] - r[1]) / 1e6:8.3f}  q{r[3] if qcol else '-'}  {name}")


if __name__ == "__main__":
    main()
