"""Per-kernel summary (calls, total, average) of a rocprofv3 --kernel-trace result database, as a markdown table.
usage: python tools/prof_summary.py results.db [steps]   (steps: divide totals by this many benchmark steps)"""
import re
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start) from kernels group by {name} "
                     "order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"| kernel | calls/step | ms/step | avg us | share |\n|---|---|---|---|---|")
    for n, cnt, tot, avg in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n)[:90]
        print(f"| `{n}` | {cnt / steps:.1f} | {tot / steps / 1e6:.3f} | {avg / 1e3:.1f} | {100 * tot / total:.1f}% |")
    print(f"\ntotal kernel time {total / steps / 1e6:.3f} ms/step over {steps:g} steps")


if __name__ == "__main__":
    main()
