#!/usr/bin/env python
"""Two-stream timeline of the last traced training step from a rocprofv3 (rocpd) database: per queue the busy time and the window it
covers, the union over queues, the idle gaps of the main queue, and the kernels that run after the main queue's last kernel.
usage: train_timeline.py <trace_results.db> [out.json]"""
import collections, json, sqlite3, sys


def union(iv):
    iv = sorted(iv)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def short(n):
    return n.replace("tnv3::", "").replace("void ", "").split("(")[0].split("<")[0][:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
    rows = list(cur.execute(f"select kernel_id, queue_id, start, end from {kd} order by start"))
    ends = [i for i, r in enumerate(rows) if "adam_multi" in names[r[0]]]
    step = rows[ends[-2] + 1: ends[-1] + 1]
    s0 = step[0][2]
    out = {"step_ms": round((step[-1][3] - s0) / 1e6, 3), "kernels": len(step)}
    qs = collections.Counter(r[1] for r in step)
    main_q = qs.most_common(1)[0][0]
    for q in qs:
        iv = [(r[2], r[3]) for r in step if r[1] == q]
        out[f"queue_{'main' if q == main_q else 'side'}_{q}"] = {
            "kernels": len(iv), "busy_ms": round(union(iv) / 1e6, 3), "first_ms": round((iv[0][0] - s0) / 1e6, 3),
            "last_end_ms": round((max(e for _, e in iv) - s0) / 1e6, 3)}
    out["union_ms"] = round(union([(r[2], r[3]) for r in step]) / 1e6, 3)
    main_iv = sorted((r[2], r[3]) for r in step if r[1] == main_q)
    main_last = max(e for _, e in main_iv[:-1])                      # the optimiser launch excluded
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(main_iv, main_iv[1:]) if b[0] - a[1] > 20e3]
    out["main_queue_gaps_over_20us"] = {"count": len(gaps), "total_ms": round(sum(gaps) / 1e3, 3), "largest_us": round(max(gaps), 1) if gaps else 0}
    tail = [r for r in step if r[1] != main_q and r[2] >= main_last]
    out["after_main_queue_done"] = {"from_ms": round((main_last - s0) / 1e6, 3), "kernels": [[short(names[r[0]]), round((r[3] - r[2]) / 1e3, 1)] for r in tail],
                                    "total_ms": round(sum(r[3] - r[2] for r in tail) / 1e6, 3)}
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        k = ("side:" if r[1] != main_q else "main:") + short(names[r[0]])
        per[k][0] += 1
        per[k][1] += (r[3] - r[2]) / 1e6
    out["per_kernel_ms"] = {k: [v[0], round(v[1], 3)] for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)


if __name__ == "__main__":
    main()
