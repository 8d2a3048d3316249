#!/usr/bin/env python3
"""GPU box: HBM traffic per unit of work from rocprofv3 PMC passes -> profiles-style JSON (gpurun_out/traffic_latest.json; copy it to
profiles/).  FETCH_SIZE and WRITE_SIZE need separate passes (MI355X_MICROARCH.md: TCC counter slots), each pass is its own run.
Per workload the command is run twice with different amounts of work (lo / hi units: iterations, traversals, repeats); the
counter totals over ALL dispatches of the run are differenced, so graph construction and warm-up cancel:
    bytes per unit = (total(hi) - total(lo)) / (units(hi) - units(lo)).
Counters are KiB (counter_defs.yaml: .../1024).  hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE: the guide's gfx950 note (FETCH_SIZE reports
half of the bytes of wide coalesced streaming reads) applied to every kernel -- exact for the streaming kernels (PageRank phases,
radix passes), an upper estimate where 4-byte gathers dominate (BFS / SSSP); fetch_kib / write_kib are kept raw beside it."""
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"


sys.path.insert(0, str(ROOT))
from bench_traversal import TRAFFIC_GROUPS, kernel_source_hash  # noqa: E402  (one definition of the hashes for collector and readers)


def source_hash():
    return kernel_source_hash()


def counter_total(cmd, counter, tag):
    d = OUT / "traffic_tmp" / tag
    subprocess.run(["rm", "-rf", str(d)])
    d.mkdir(parents=True, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", str(d), "-o", "run", "--"] + cmd
    r = subprocess.run(full, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    dbs = glob.glob(str(d / "**" / "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        raise RuntimeError(f"rocprofv3 failed ({r.returncode}) for {tag}: {r.stdout[-800:]}")
    tot, n = 0.0, 0
    for db in dbs:
        c = sqlite3.connect(db)
        row = c.execute("select sum(value), count(*) from counters_collection where counter_name = ?", (counter,)).fetchone()
        tot += row[0] or 0.0
        n += row[1] or 0
        c.close()
        os.remove(db)
    return tot, n


def measure(key, cmd_lo, cmd_hi, units_lo, units_hi, what):
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        lo, n_lo = counter_total(cmd_lo, counter, f"{key}_{counter}_lo")
        hi, n_hi = counter_total(cmd_hi, counter, f"{key}_{counter}_hi")
        res[counter] = (hi - lo) / (units_hi - units_lo)
        res[counter + "_dispatches"] = (n_lo, n_hi)
    return {"hbm_bytes": int((2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024), "fetch_kib": round(res["FETCH_SIZE"], 1), "write_kib": round(res["WRITE_SIZE"], 1),
            "unit": what, "units": [units_lo, units_hi], "dispatches": [res["FETCH_SIZE_dispatches"], res["WRITE_SIZE_dispatches"]]}


def main():
    py = sys.executable
    only = set(sys.argv[1:])
    bench = [py, str(ROOT / "bench.py"), "--scale", "26", "--warmup", "1", "--no-check", "--no-cpu-baseline", "--no-extras", "--no-phase-pass", "--placements", "1"]
    trav = [py, str(ROOT / "bench_traversal.py"), "--scale", "24", "--no-cpu-baseline", "--no-check", "--single-variant"]  # with predecessors (the headline since round 5)
    louv = [py, str(ROOT / "bench_louvain.py"), "--scale", "22", "--cpu-scale", "0"]
    work = {
        "pagerank_s26": (bench + ["--steps", "4"], bench + ["--steps", "24"], 4, 24, "one power iteration (k_tiled_phase1 + k_tiled_phase2)"),
        "bfs_s24_int_pred": (trav + ["--weights", "int", "--no-sssp", "--roots", "2"], trav + ["--weights", "int", "--no-sssp", "--roots", "10"], 2, 10, "one BFS with predecessors (all levels)"),
        "sssp_s24_int_pred": (trav + ["--weights", "int", "--roots", "2"], trav + ["--weights", "int", "--roots", "6"], 2, 6, "one BFS + one SSSP (subtract bfs_s24_int_pred)"),
        "sssp_s24_unit_pred": (trav + ["--weights", "unit", "--roots", "2"], trav + ["--weights", "unit", "--roots", "6"], 2, 6, "one BFS + one SSSP (subtract bfs_s24_int_pred)"),
        "louvain_s22": (louv + ["--repeats", "1"], louv + ["--repeats", "4"], 1, 4, "one cugraph_louvain call (all levels)"),
    }
    entries = {}
    for key, (lo, hi, ulo, uhi, what) in work.items():
        if only and key not in only:
            continue
        try:
            entries[key] = measure(key, lo, hi, ulo, uhi, what)
        except Exception as e:
            entries[key] = {"error": repr(e)[:400]}
        print(key, entries[key], flush=True)
    for k in ("sssp_s24_int_pred", "sssp_s24_unit_pred"):  # the traversal bench runs one BFS and one SSSP per root: isolate the SSSP
        if k in entries and "hbm_bytes" in entries[k] and "hbm_bytes" in entries.get("bfs_s24_int_pred", {}):
            b = entries["bfs_s24_int_pred"]
            entries[k] = dict(entries[k], hbm_bytes=entries[k]["hbm_bytes"] - b["hbm_bytes"], fetch_kib=round(entries[k]["fetch_kib"] - b["fetch_kib"], 1),
                              write_kib=round(entries[k]["write_kib"] - b["write_kib"], 1), unit="one SSSP with predecessors (all rounds)")
    out = {"source_hash": source_hash(), "group_hashes": {g: kernel_source_hash(g) for g in TRAFFIC_GROUPS}, "groups": TRAFFIC_GROUPS, "source": "tools/traffic_collect.py on the GPU box: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), totals over all dispatches, "
                                                   "differenced between two amounts of work; KiB x 1024; FETCH_SIZE x 2 per MI355X_MICROARCH.md (gfx950)",
           "entries": entries}
    (OUT / "traffic_latest.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out)[:300])


if __name__ == "__main__":
    main()
