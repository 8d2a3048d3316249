"""The bench line the driver consumes: the committed line of the last measured run (profiles/) must carry every field of the
contract, and its derived numbers must be consistent with each other."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def latest_bench_line():
    files = sorted((ROOT / "profiles").glob("r*_bench_s26.json"))
    assert files, "no committed bench line under profiles/"
    return json.loads(files[-1].read_text().strip().splitlines()[-1]), files[-1].name


def test_bench_line_contract():
    d, name = latest_bench_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, (name, key)
    assert d["metric"] == "pagerank_mteps_rmat26" and d["unit"] == "MTEPS" and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "RMAT scale 26" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    ne, nv = d["config"]["edges"], d["config"]["vertices"]
    assert r["algorithmic_bytes_per_launch"] == 4 * ne + 16 * nv + 4
    if "frac_basis" in r:  # round 5: priced on the wall clock of the timed region (the two kernels of consecutive iterations may overlap)
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
        if r.get("region_event_ms_per_step") is not None:  # round 6: one event pair around the timed steps; the phase averages come from a second pass
            assert r["region_event_ms_per_step"] <= d["ms_per_step"] * 1.001      # the stream's own clock cannot exceed the host's around it
            assert r["avg_kernel_ms"] <= d["ms_per_step"] * 1.05                  # another pass of the same steps: run-to-run spread only
        elif r.get("frac_kernels") is not None:  # round 5: events inside the timed pass: the kernels' own time cannot exceed the step
            assert r["frac_kernels"] >= r["frac"] - 1e-3 and r["avg_kernel_ms"] <= d["ms_per_step"] * 1.001
    else:
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert abs(d["value"] - ne / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3   # MTEPS = E * iterations / time
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]   # real HBM bytes cannot undercut the algorithmic ones
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1


def test_extras_carry_configs_3_and_5():
    """Round 3: BASELINE configs 3 (BFS + SSSP at RMAT-24) and 5's graph size for one GPU (Louvain at RMAT-22) ride in the driver's line
    as sub-objects of the same shape: value / unit, roofline (bound, achieved, peak, frac, traffic), check, cpu_baseline."""
    d, name = latest_bench_line()
    if "extra" not in d:  # lines of rounds 1-2
        return
    e = d["extra"]
    for key in ("bfs", "sssp", "sssp_unit", "louvain"):
        assert key in e, (name, key, [k for k in e if k.endswith("_error")])
        x = e[key]
        for f in ("value", "unit", "roofline", "check"):
            assert f in x, (key, f)
        r = x["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
        assert x["check"]["ok"] is True
    for key in ("bfs", "sssp", "louvain"):
        c = e[key]["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and "sample" in c
    assert e["bfs"]["metric"] == "bfs_mteps_rmat24" and e["sssp"]["unit"] == "MTEPS" and e["louvain"]["unit"] == "s"
    assert e["sssp_unit"].get("unit_weight_distances_equal_bfs") is True  # integer hops == BFS distances, bit for bit


def test_traffic_file_matches_bench_line():
    """profiles/traffic_latest.json (tools/traffic_collect.py): hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per unit of work, keyed by
    the hash of the kernels' sources; the committed bench line quotes exactly that figure (or null with a reason when the sources moved on)."""
    t = json.loads((ROOT / "profiles" / "traffic_latest.json").read_text())
    assert "source_hash" in t and "entries" in t
    pr = t["entries"]["pagerank_s26"]
    assert abs(pr["hbm_bytes"] - (2 * pr["fetch_kib"] + pr["write_kib"]) * 1024) <= 2048
    d, _ = latest_bench_line()
    r = d["roofline"]
    if r["traffic"] is not None:
        assert r["traffic"] == pr["hbm_bytes"] and "same source hash" in r["traffic_source"]
        assert r["traffic"] >= r["algorithmic_bytes_per_launch"]
    else:
        assert "STALE" in r["traffic_source"] or "absent" in r["traffic_source"] or "no entry" in r["traffic_source"]


def test_traffic_hash_groups_cover_the_sources(tmp_path, monkeypatch):
    """The counter file is keyed per workload by a hash over the sources that workload's kernels come from (bench_traversal.TRAFFIC_GROUPS).
    Every file of csrc/ must be in a group or on the explicit list of files no measured unit of work runs -- a new source file cannot
    silently escape the staleness check -- and an edit inside a group must turn exactly that group's figures stale."""
    import sys

    sys.path.insert(0, str(ROOT))
    import bench_traversal as bt

    files = {f.name for f in (ROOT / "cugraph_amd" / "csrc").glob("*.h*")}
    grouped = set().union(*bt.TRAFFIC_GROUPS.values())
    assert grouped <= files and set(bt.TRAFFIC_UNMEASURED) <= files
    assert files == grouped | set(bt.TRAFFIC_UNMEASURED), files ^ (grouped | set(bt.TRAFFIC_UNMEASURED))
    assert not (grouped & set(bt.TRAFFIC_UNMEASURED))
    t = json.loads((ROOT / "profiles" / "traffic_latest.json").read_text())
    assert set(t["group_hashes"]) == set(bt.TRAFFIC_GROUPS)
    # a copy of the tree with one traversal source edited: PageRank's figure stays valid, the traversal figures go stale
    import shutil

    root2 = tmp_path / "repo"
    (root2 / "cugraph_amd").mkdir(parents=True)
    shutil.copytree(ROOT / "cugraph_amd" / "csrc", root2 / "cugraph_amd" / "csrc")
    (root2 / "profiles").mkdir()
    shutil.copy(ROOT / "profiles" / "traffic_latest.json", root2 / "profiles" / "traffic_latest.json")
    monkeypatch.setattr(bt, "ROOT", root2)
    fresh = t["group_hashes"]["pagerank"] == bt.kernel_source_hash("pagerank")  # (false while the sources are ahead of the committed collection)
    with open(root2 / "cugraph_amd" / "csrc" / "traversal.hip", "a") as f:
        f.write("// edited\n")
    pr, why_pr = bt.counter_traffic("pagerank_s26")
    bfs, why_bfs = bt.counter_traffic("bfs_s24_int")
    assert bfs is None and "STALE" in why_bfs
    assert (pr is not None) == fresh
