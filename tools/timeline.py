"""Timeline of the last step of a rocprofv3 --kernel-trace run (rocpd sqlite output): every kernel with its start relative to the
step's first kernel, its duration and the idle gap in front of it, per stream; the sum of the gaps on the main stream.

usage: python tools/timeline.py <dir-with-.db> [first-kernel-substring (default lz77_match7)]"""
import glob, os, sqlite3, sys


def main():
    root = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "lz77_match7"
    db = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = sorted(con.execute(f"SELECT {s}, {e}, {name}, {q} FROM kernels"))
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    if len(starts) < 2:
        sys.exit("fewer than two steps in the trace")
    a, b = starts[-2], starts[-1]          # the last complete step
    # the step begins a little earlier: fills / uploads in front of the first match kernel
    while a > 0 and rows[a][0] - rows[a - 1][1] < 30000 and first not in rows[a - 1][2] and "checksum" not in rows[a - 1][2] and "materialize" not in rows[a - 1][2]:
        a -= 1
    step = rows[a:b]
    t0 = step[0][0]
    busy_end = t0
    gaps = 0
    print("start_us,dur_us,gap_us,queue,kernel")
    for st, en, n, qu in step:
        gap = st - busy_end
        if gap > 0:
            gaps += gap
        print("%.1f,%.1f,%.1f,%s,%s" % ((st - t0) / 1e3, (en - st) / 1e3, gap / 1e3, qu, n.split("(")[0][:60]))
        busy_end = max(busy_end, en)
    print("# step span %.1f us, idle (no kernel running on any queue) %.1f us" % ((busy_end - t0) / 1e3, gaps / 1e3))


if __name__ == "__main__":
    main()
