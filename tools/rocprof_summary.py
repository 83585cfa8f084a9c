#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db: per-kernel count / avg / min / total (us)."""
import glob
import sqlite3
import sys


def main(path):
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    for db in dbs:
        c = sqlite3.connect(db)
        names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        sym = [n for n in names if n.startswith("rocpd_info_kernel_symbol")][0]
        dis = [n for n in names if n.startswith("rocpd_kernel_dispatch")][0]
        q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, "
             f"sum(d.end-d.start)/1000.0 from {dis} d join {sym} s on d.kernel_id=s.id "
             f"group by s.kernel_name order by 5 desc")
        print("# %s" % db)
        print("%-110s %6s %10s %10s %12s" % ("kernel", "calls", "avg_us", "min_us", "total_us"))
        for name, n, avg, mn, tot in c.execute(q):
            print("%-110s %6d %10.1f %10.1f %12.1f" % (name[:110], n, avg, mn, tot))


if __name__ == "__main__":
    main(sys.argv[1])
