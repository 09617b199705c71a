"""Matrix-pipe occupancy per kernel from a rocprofv3 --pmc results database holding SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE (summed over the 8 XCDs; / 8 = the kernel's cycles; MFMA_BUSY is summed over the 1024 SIMDs):
busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024). Usage: python tools/mfma_busy_md.py results.db"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, avg(value), count(*) from counters_collection "
                     f"group by {name_col}, counter_name").fetchall()
    per = {}
    for kname, counter, avg, n in rows:
        per.setdefault(kname, {})[counter] = (avg, n)
    out = []
    for kname, d in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        busy, n = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        gui = d["GRBM_GUI_ACTIVE"][0]
        cyc = gui / 8.0
        frac = busy / (cyc * 1024.0) if cyc > 0 else 0.0
        if frac < 0.003:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", kname)
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)[:80]
        out.append((cyc * n, name, n, cyc, frac))
    print("| kernel | dispatches | kernel cycles (avg) | MFMA busy |\n|---|---|---|---|")
    for _, name, n, cyc, frac in sorted(out, reverse=True):
        print(f"| `{name}` | {n} | {cyc:.0f} | {100 * frac:.1f} % |")


if __name__ == "__main__":
    main()
