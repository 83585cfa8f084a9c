#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes as
MI355X_MICROARCH.md prescribes) of `bench.py` into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic.json

Units / corrections: both counters are in KB (x1024); on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read, so fetch bytes are doubled (calibrated here on
the strided pass, which reads exactly batch*N*8 B = 512 MiB: FETCH_SIZE x 2 = 512.2 MiB)."""
import glob
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [n for n in names if n.startswith(p)][0]  # noqa: E731
    q = (f"select s.kernel_name, count(*), avg(p.value) from {t('rocpd_pmc_event')} p "
         f"join {t('rocpd_info_pmc')} i on p.pmc_id=i.id "
         f"join {t('rocpd_kernel_dispatch')} d on p.event_id=d.event_id "
         f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id=s.id "
         f"where i.name='{counter}' group by s.kernel_name")
    return {r[0]: (r[1], r[2]) for r in c.execute(q)}


def main(fetch_dir, write_dir):
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    out = {"unit": "bytes per launch", "fetch_correction": 2.0, "kernels": {}}
    total = 0.0
    for k in f:
        fb = f[k][1] * 1024 * 2.0
        wb = w.get(k, (0, 0.0))[1] * 1024
        out["kernels"][k] = {"launches": f[k][0], "fetch_bytes": fb, "write_bytes": wb,
                             "raw_FETCH_SIZE_KB": f[k][1], "raw_WRITE_SIZE_KB": w.get(k, (0, 0.0))[1]}
        total += fb + wb
    out["bytes_per_call"] = total
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
