#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories:
    python tools/pmc_dump.py <dir> [<dir> ...]"""
import glob
import sqlite3
import sys


def dump(path):
    for db in glob.glob(path + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        t = lambda p: [n for n in names if n.startswith(p)][0]  # noqa: E731
        q = (f"select s.kernel_name, i.name, count(*), avg(p.value) from {t('rocpd_pmc_event')} p "
             f"join {t('rocpd_info_pmc')} i on p.pmc_id=i.id "
             f"join {t('rocpd_kernel_dispatch')} d on p.event_id=d.event_id "
             f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id=s.id "
             f"group by s.kernel_name, i.name order by s.kernel_name, i.name")
        for k, n, cnt, avg in c.execute(q):
            print("%-28s %6d %16.1f  %s" % (n, cnt, avg, k[:110]))


for p in sys.argv[1:]:
    dump(p)
