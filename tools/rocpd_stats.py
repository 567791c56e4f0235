"""Kernel statistics from a rocprofv3 rocpd database (this ROCm's default output):  python tools/rocpd_stats.py <results.db> [last_fraction]
-> name, calls, total us, avg us, share -- optionally over the LAST fraction of the trace only (the timed part of a bench run)."""
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    t0, t1 = cur.execute("select min(start), max(end) from %s" % kd).fetchone()
    cut = t1 - (t1 - t0) * frac
    rows = cur.execute("select s.kernel_name, count(*), sum(d.end - d.start), min(d.start), max(d.end) from %s d join %s s on d.kernel_id = s.id "
                       "where d.start >= ? group by s.kernel_name order by 3 desc" % (kd, ks), (cut,)).fetchall()
    total = sum(r[2] for r in rows)
    span = max(r[4] for r in rows) - min(r[3] for r in rows)
    print("window %.3f ms, kernels busy %.3f ms (%.0f %%), %d launches" % (span / 1e6, total / 1e6, 100.0 * total / span, sum(r[1] for r in rows)))
    print("%-90s %8s %12s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for name, n, tot, _, _ in rows[:40]:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)
        print("%-90s %8d %12.1f %9.2f %6.1f" % (short[:90], n, tot / 1e3, tot / 1e3 / n, 100.0 * tot / total))


if __name__ == "__main__":
    main()
