"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite output) as CSV: one row per kernel.

usage: python tools/prof_summary.py <dir-with-.db> [header comment ...] > profiles/<name>.csv
"""
import glob, os, sqlite3, sys


def main():
    root = sys.argv[1]
    dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
    if not dbs:
        sys.exit("no .db under " + root)
    rows = {}
    for db in dbs:
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
        name = "name" if "name" in cols else "kernel_name"
        s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        for n, a, b in con.execute(f"SELECT {name}, {s}, {e} FROM kernels"):
            rows.setdefault(n, []).append(b - a)
    total = sum(sum(v) for v in rows.values())
    for c in sys.argv[2:]:
        print("# " + c)
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        short = n.split("(")[0]
        print(f'"{short}",{len(v)},{sum(v)},{sum(v) // len(v)},{min(v)},{max(v)},{100.0 * sum(v) / total:.2f}')


if __name__ == "__main__":
    main()
