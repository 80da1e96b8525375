#!/usr/bin/env python
"""Per-kernel sums of PMC counters from a rocprofv3 rocpd sqlite database.
usage: rocpd_pmc.py results.db [out.csv]   (prints the schema of the pmc views when they are empty/unknown)"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
rows = []
try:
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    if cols:
        namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
        ccol = "counter_name" if "counter_name" in cols else None
        vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        if ccol and vcol:
            rows = db.execute(f"select {namecol}, {ccol}, count(*), sum({vcol}), avg({vcol}) from counters_collection "
                              f"group by {namecol}, {ccol} order by sum({vcol}) desc").fetchall()
        else:
            print("counters_collection columns:", cols)
except sqlite3.Error as e:
    print("pmc query failed:", e)
if not rows:
    for v in views:
        if "pmc" in v or "counter" in v:
            print(v, [r[1] for r in db.execute(f"pragma table_info('{v}')")], db.execute(f"select count(*) from '{v}'").fetchone())
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(("kernel", "counter", "dispatches", "sum", "avg_per_dispatch"))
w.writerows(rows)
