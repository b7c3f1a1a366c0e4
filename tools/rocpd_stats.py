#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into the same table `--stats` prints:
per-kernel calls / total / average / min / max / share.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
    name = name.replace("void ", "").replace("sbr::", "")
    name = re.sub(r"rocprim::[a-z_:]*detail::", "rocprim::", name)
    return name[:100]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
