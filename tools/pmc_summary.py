"""Average a rocprofv3 --pmc counter per kernel from the *_counter_collection.csv files under a directory.

usage: python tools/pmc_summary.py <dir> [comment ...]
"""
import csv, glob, os, sys


def main():
    root = sys.argv[1]
    files = sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True))
    if not files:
        sys.exit("no *_counter_collection.csv under " + root)
    acc = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name")
                cname = row.get("Counter_Name") or row.get("counter_name")
                val = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                did = row.get("Dispatch_Id") or row.get("dispatch_id")
                acc.setdefault((name.split("(")[0], cname), {}).setdefault(did, 0.0)
                acc[(name.split("(")[0], cname)][did] += val   # rows are per XCD / dimension: sum per dispatch
    for c in sys.argv[2:]:
        print("# " + c)
    print("Name,Counter,Dispatches,AveragePerDispatch")
    for (name, cname), d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        print(f'"{name}",{cname},{len(d)},{sum(d.values()) / len(d):.1f}')


if __name__ == "__main__":
    main()
