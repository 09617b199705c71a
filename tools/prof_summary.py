"""Summarise a rocprofv3 --kernel-trace --stats results database as a markdown table (per-kernel calls, total,
average, share) plus per-step totals.  Usage: python tools/prof_summary.py results.db STEPS_IN_RUN "title"."""
import sqlite3
import sys


def main():
    db, steps, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# {title}\n")
    print(f"GPU kernel time: {total / 1e6:.2f} ms over {steps} steps = {total / 1e6 / steps:.3f} ms/step\n")
    print("| kernel | calls | calls/step | total ms | avg us | ms/step | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg in rows:
        if tot / total < 0.0005:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short if len(short) <= 90 else short[:87] + "..."
        print(f"| {short} | {n} | {n / steps:.1f} | {tot / 1e6:.2f} | {avg / 1e3:.1f} | {tot / 1e6 / steps:.3f} | "
              f"{100 * tot / total:.1f} |")


if __name__ == "__main__":
    main()
