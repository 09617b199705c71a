"""Average the PMC counters of a rocprofv3 --pmc results database per kernel name.
Usage: python tools/pmc_summary.py results.db [kernel-substring]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, avg(value), count(*) from counters_collection "
                     f"group by {name_col}, counter_name").fetchall()
    for kname, counter, avg, n in rows:
        if sub in kname:
            print(f"{kname[:70]:70s} {counter:32s} {avg:16.1f}  (n={n})")


if __name__ == "__main__":
    main()
