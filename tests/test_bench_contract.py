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
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert abs(d["value"] - ne / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3   # MTEPS = E * iterations / time
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]   # real HBM bytes cannot undercut the algorithmic ones
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1


def test_traffic_file_matches_profile_summary():
    t = json.loads((ROOT / "profiles" / "traffic_latest.json").read_text())
    src = t["source"].split(" ")[0]
    assert (ROOT / src).exists(), src
    p = t["per_iteration"]
    total = 2 * (p["k_tiled_phase1"]["FETCH_SIZE_KB"] + p["k_tiled_phase2"]["FETCH_SIZE_KB"]) + p["k_tiled_phase1"]["WRITE_SIZE_KB"] + \
        p["k_tiled_phase2"]["WRITE_SIZE_KB"]
    assert abs(t["hbm_bytes_per_launch"] - total * 1000) <= 1000
    d, _ = latest_bench_line()
    assert d["roofline"]["traffic"] == t["hbm_bytes_per_launch"]
