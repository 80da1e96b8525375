#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: rocpd_summary.py trace_results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start),
                            max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x)
                     from kernels group by name order by sum(end-start) desc""").fetchall()
total = sum(r[2] for r in rows)
out = [("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "arch_vgpr", "accum_vgpr", "sgpr", "lds_bytes", "max_grid_x", "wg_x")]
for r in rows:
    out.append((r[0], r[1], round(r[2] / 1e6, 4), round(r[3] / 1e3, 2), round(r[4] / 1e3, 2), round(r[5] / 1e3, 2),
                round(100.0 * r[2] / total, 2), r[6], r[7], r[8], r[9], r[10], r[11]))
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
