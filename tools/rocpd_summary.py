#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs: per-kernel time stats and per-kernel PMC counter averages.
usage: rocpd_summary.py <dir-or-db> [...]  -> prints a text table (commit the output under profiles/)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def summarize(db):
    c = sqlite3.connect(db)
    cur = c.cursor()
    print(f"# {db}")
    try:
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
        for n, k, s, a, mn, mx in rows[:25]:
            print(f"{n[:70]:<70} {k:>6} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*s/tot:>6.1f}")
    except Exception as e:
        print("kernels view:", e)
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if cols:
            kn = "kernel_name" if "kernel_name" in cols else "name"
            rows = cur.execute(f"select {kn}, counter_name, avg(value), count(*) from counters_collection group by {kn}, counter_name").fetchall()
            agg = defaultdict(dict)
            for k, cn, v, n in rows:
                agg[k][cn] = (v, n)
            for k, d in agg.items():
                if any(t in k for t in ("spmv", "tiled", "bfs", "sssp", "advance")):
                    print(f"  PMC {k[:80]}")
                    for cn, (v, n) in sorted(d.items()):
                        print(f"      {cn:<40} avg/dispatch = {v:,.1f}   (n={n})")
    except Exception as e:
        print("counters_collection:", e)


def per_dispatch(db, regex):
    """one line per dispatch of the kernels matching `regex`: duration (when the kernels view has it) and every counter"""
    import re

    c = sqlite3.connect(db)
    cur = c.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if not cols:
        return
    kn = "kernel_name" if "kernel_name" in cols else "name"
    did = next((x for x in ("dispatch_id", "id", "kernel_id") if x in cols), None)
    if did is None:
        print("per-dispatch: no dispatch id column in", cols)
        return
    rows = cur.execute(f"select {did}, {kn}, counter_name, sum(value) from counters_collection group by {did}, {kn}, counter_name order by {did}").fetchall()
    dur = {}
    try:
        kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        kd = next((x for x in ("dispatch_id", "id") if x in kcols), None)
        if kd:
            dur = {r[0]: r[1] for r in cur.execute(f"select {kd}, duration from kernels")}
    except Exception:
        pass
    table = defaultdict(dict)
    names = {}
    for d, k, cn, v in rows:
        if re.search(regex, k):
            table[d][cn] = v
            names[d] = k
    if not table:
        return
    cn_all = sorted({cn for d in table.values() for cn in d})
    print(f"# per dispatch ({regex}) {db}")
    print("dispatch  dur_us " + " ".join(f"{cn:>22}" for cn in cn_all))
    for d in sorted(table):
        print(f"{d:>8} {dur.get(d, 0) / 1e3:>7.1f} " + " ".join(f"{table[d].get(cn, 0):>22,.0f}" for cn in cn_all))


if __name__ == "__main__":
    if "--per-dispatch" in sys.argv:
        i = sys.argv.index("--per-dispatch")
        rx = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        for a in sys.argv[1:]:
            dbs = [a] if a.endswith(".db") else sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
            for d in dbs:
                try:
                    per_dispatch(d, rx)
                except Exception as e:
                    print("per-dispatch:", d, e)
        sys.exit(0)
    for a in sys.argv[1:]:
        dbs = [a] if a.endswith(".db") else sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
        for d in dbs:
            summarize(d)
