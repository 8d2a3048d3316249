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
        # the library's kernels live in namespace cga; everything else in a bench process is the harness (torch / rocprim sorts that build the
        # input, the post-timing checks, runtime copies): one line at the end, not rows among the library's
        lib = [r for r in rows if "cga::" in r[0]]
        other = [r for r in rows if "cga::" not in r[0]]
        if not lib:
            lib, other = rows, []
        tot = sum(r[2] for r in lib) or 1
        print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
        for n, k, s, a, mn, mx in lib[:25]:
            print(f"{n[:70]:<70} {k:>6} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*s/tot:>6.1f}")
        if other:
            print(f"(pct = share of the library's kernel time, {tot / 1e6:.3f} ms; harness kernels outside namespace cga -- torch / rocprim / runtime copies and fills: "
                  f"{sum(r[1] for r in other)} dispatches, {sum(r[2] for r in other) / 1e6:.3f} ms)")
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


def last_dispatches(db, n, regex):
    """stats of the LAST n dispatches of every kernel matching `regex` (by start time): with `bench.py --no-phase-pass` these are the launches of the timed
    region -- a tuned plan's earlier launches ran on the placements its tuning tried and discarded"""
    import re

    c = sqlite3.connect(db)
    cur = c.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    order = next((x for x in ("start", "start_time", "begin", "id") if x in cols), None)
    if order is None:
        print("last dispatches: no start column in", cols)
        return
    names = [r[0] for r in cur.execute("select distinct name from kernels")]
    print(f"# last {n} dispatches per kernel matching /{regex}/ (ordered by {order})")
    tot = 0.0
    for name in names:
        if not re.search(regex, name):
            continue
        d = [r[0] for r in cur.execute(f"select duration from kernels where name = ? order by {order} desc limit ?", (name, n))]
        if d:
            tot += sum(d) / len(d)
            print(f"{name[:70]:<70} {len(d):>6} {sum(d)/1e6:>10.3f} {sum(d)/len(d)/1e3:>10.2f} {min(d)/1e3:>10.2f} {max(d)/1e3:>10.2f}")
    print(f"# sum of the averages: {tot / 1e3:.2f} us")


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


def overlap(db, rx_a, rx_b, show=12):
    """Concurrency of two kernel families in a kernel trace: total time of A, of B, and the time during which a dispatch of A and a
    dispatch of B were both running (start / end timestamps of the `kernels` view); plus the first `show` dispatches as a timeline."""
    import re

    c = sqlite3.connect(db)
    cur = c.cursor()
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    st = next((x for x in ("start", "start_timestamp", "begin") if x in kcols), None)
    en = next((x for x in ("end", "end_timestamp") if x in kcols), None)
    if st is None or en is None:
        print("overlap: no start/end columns in", kcols)
        return
    rows = cur.execute(f"select name, {st}, {en} from kernels order by {st}").fetchall()
    a = [(s0, e0) for n, s0, e0 in rows if re.search(rx_a, n)]
    b = [(s0, e0) for n, s0, e0 in rows if re.search(rx_b, n)]
    if not a or not b:
        print(f"overlap: {len(a)} dispatches of /{rx_a}/, {len(b)} of /{rx_b}/ in {db}")
        return
    both, j = 0, 0
    for s0, e0 in a:  # both lists are sorted by start; dispatches of one family do not overlap each other (one stream each)
        while j < len(b) and b[j][1] <= s0:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e0:
            both += max(0, min(e0, b[k][1]) - max(s0, b[k][0]))
            k += 1
    ta, tb = sum(e0 - s0 for s0, e0 in a), sum(e0 - s0 for s0, e0 in b)
    span = max(a[-1][1], b[-1][1]) - min(a[0][0], b[0][0])
    print(f"# overlap {db}")
    print(f"/{rx_a}/: {len(a)} dispatches, {ta / 1e6:.3f} ms   /{rx_b}/: {len(b)} dispatches, {tb / 1e6:.3f} ms   both running: {both / 1e6:.3f} ms "
          f"({100 * both / max(tb, 1):.1f} % of B)   first start -> last end: {span / 1e6:.3f} ms   sum A + B: {(ta + tb) / 1e6:.3f} ms")
    t0 = min(a[0][0], b[0][0])
    ev = sorted([(s0, e0, "A") for s0, e0 in a[:show]] + [(s0, e0, "B") for s0, e0 in b[:show]])
    for s0, e0, w in ev:
        print(f"   {w}  start {(s0 - t0) / 1e3:>10.1f} us  end {(e0 - t0) / 1e3:>10.1f} us  dur {(e0 - s0) / 1e3:>8.1f} us")


def segments(db, marker, last=6, top=9):
    """Kernel time between consecutive dispatches of the kernel matching `marker` (Louvain: k_lv_chunk_prep runs once per level, so a
    segment = one level + the contraction behind it): the `last` segments of the trace, the `top` kernels of each."""
    import re

    c = sqlite3.connect(db)
    cur = c.cursor()
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    st = next((x for x in ("start", "start_timestamp", "begin") if x in kcols), None)
    en = next((x for x in ("end", "end_timestamp") if x in kcols), None)
    if st is None or en is None:
        print("segments: no start/end columns in", kcols)
        return
    rows = cur.execute(f"select name, {st}, {en} from kernels order by {st}").fetchall()
    cuts = [i for i, r in enumerate(rows) if re.search(marker, r[0])]
    if not cuts:
        print(f"segments: no dispatch matches /{marker}/ in {db}")
        return
    cuts.append(len(rows))
    print(f"# segments between dispatches of /{marker}/ (the last {last}) {db}")
    for a, b in list(zip(cuts[:-1], cuts[1:]))[-last:]:
        seg = [r for r in rows[a:b] if "cga::" in r[0]]
        if not seg:
            continue
        agg = defaultdict(lambda: [0, 0])
        for n, s0, e0 in seg:
            k = re.sub(r"\(.*", "", n.replace("cga::(anonymous namespace)::", "").replace("void ", ""))
            agg[k][0] += 1
            agg[k][1] += e0 - s0
        busy = sum(v[1] for v in agg.values())
        print(f"  segment of {len(seg)} dispatches: wall {(seg[-1][2] - seg[0][1]) / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms: "
              + ", ".join(f"{k} {v[0]}x {v[1] / 1e6:.3f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]))


if __name__ == "__main__":
    if "--overlap" in sys.argv:
        i = sys.argv.index("--overlap")
        rx_a, rx_b = sys.argv[i + 1], sys.argv[i + 2]
        del sys.argv[i:i + 3]
        for a in sys.argv[1:]:
            dbs = [a] if a.endswith(".db") else sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
            for d in dbs:
                try:
                    overlap(d, rx_a, rx_b)
                except Exception as e:
                    print("overlap:", d, e)
        sys.exit(0)
    if "--segments" in sys.argv:
        i = sys.argv.index("--segments")
        rx = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        for a in sys.argv[1:]:
            dbs = [a] if a.endswith(".db") else sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
            for d in dbs:
                try:
                    segments(d, rx)
                except Exception as e:
                    print("segments:", d, e)
        sys.exit(0)
    last = None
    if "--last" in sys.argv:  # --last N REGEX: after the summary, the last N dispatches of the matching kernels
        i = sys.argv.index("--last")
        last = (int(sys.argv[i + 1]), sys.argv[i + 2])
        del sys.argv[i:i + 3]
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
            if last:
                try:
                    last_dispatches(d, last[0], last[1])
                except Exception as e:
                    print("last:", d, e)
