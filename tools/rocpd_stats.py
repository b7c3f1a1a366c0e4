#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into the same table `--stats` prints:
per-kernel calls / total / average / min / max / share.  Usage: rocpd_stats.py results.db [out.md] [--window KERNEL FIRST COUNT]
--window: a second table over the TIMED region only — the dispatches from the FIRST-th dispatch of KERNEL (a substring; in start order)
up to, not including, its (FIRST + COUNT)-th: what bench.py's HIP events bracket (warm-up and the second, all-timers pass left out)."""
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
    if "--window" in sys.argv:
        i = sys.argv.index("--window")
        kern, first, count = sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        if "start" in cols:
            allr = db.execute("select name, start, duration from kernels order by start").fetchall()
            marks = [j for j, r in enumerate(allr) if kern in r[0]]
            if len(marks) >= first + count:
                win = allr[marks[first]:marks[first + count] if len(marks) > first + count else len(allr)]
                agg = {}
                for name, _, dur in win:
                    a = agg.setdefault(name, [0, 0, None, 0])
                    a[0] += 1; a[1] += dur; a[2] = dur if a[2] is None else min(a[2], dur); a[3] = max(a[3], dur)
                tot2 = sum(a[1] for a in agg.values()) or 1
                lines2 = ["", f"Timed region only: the dispatches from dispatch {first} of `{kern}` up to its dispatch {first + count} ({count} steps).", "",
                          "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
                for name, (n, tot, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    lines2.append(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / tot2:.1f} |")
                text += "\n" + "\n".join(lines2)
    print(text)
    out = [a for a in sys.argv[2:3] if not a.startswith("--")]
    if out:
        open(out[0], "w").write(text + "\n")


if __name__ == "__main__":
    main()
