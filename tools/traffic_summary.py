"""HBM-side bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the
same command), as a markdown table and a JSON file. On gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-
lane reads at 64 bytes, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact. Counter values are KB per dispatch.
usage: python tools/traffic_summary.py fetch.db write.db out.json > out.md"""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, avg(value), count(*) from counters_collection where counter_name = ? "
                     f"group by {name_col}", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n)[:80]


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k, (f, n) in fetch.items():
        w = write.get(k, (0.0, 0))[0]
        rows.append((short(k), n, f, 2 * f / 1e3, w / 1e3, (2 * f + w) / 1e3))
    rows.sort(key=lambda r: -r[5] * r[1])
    print("| kernel | dispatches | FETCH_SIZE KB | read MB (x2) | write MB | total MB per launch |\n|---|---|---|---|---|---|")
    for r in rows[:28]:
        print(f"| `{r[0]}` | {r[1]} | {r[2]:.0f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} |")
    conv = [r for r in rows if r[0].startswith(("void conv_patch_kernel", "void conv_igemm_kernel", "conv1x1_kernel"))]
    launches = sum(r[1] for r in conv)
    out = {"conv_fwd_dgrad_bytes_per_launch": sum(r[5] * 1e6 * r[1] for r in conv) / max(launches, 1),
           "per_kernel_MB": {r[0]: {"dispatches": r[1], "read": round(r[3], 2), "write": round(r[4], 2)} for r in rows[:28]}}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
